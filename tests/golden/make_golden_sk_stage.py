#!/usr/bin/env python3
"""Golden fixture of the reference's DEFORM, produced by running the UNMODIFIED ``networks/sk_gs.py`` on CPU.

Run in the build container only (needs /root/reference, read-only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sk_stage.py            (--check: compare with the committed file instead)
Only ``sk_stage.npz`` travels.  Third-party modules this image lacks are the inert stubs of make_golden.py -- EXCEPT the two the
deform needs, which are this repository's stand-ins, installed the way INTEGRATION.md tells a user of the reference to:

    sk_gs_amd.install_as_lietorch()      # `from lietorch import SE3, SO3`         (networks/sk_gs.py:12)
    sk_gs_amd.install_as_pytorch3d()     # `from pytorch3d.ops import knn_points`  (networks/sk_gs.py:11)

What runs is the reference's own class: ``SkeletonGaussianSplatting(...)`` is constructed with the shipped configs' options
(exps/default.yaml:20-37, d_nerf_sc_gs.yaml:31-32, d_nerf_sp_gs.yaml:30-32), its parameters are filled from a seeded generator, and
``model.forward(t=..., stage='sk' | 'sp', time_id=...)`` (sk_gs.py:1160-1204 -> sk_stage :1109-1150 -> kinematic :1069-1107 ->
skeleton_warp_SE3 :193-206 -> calc_LBS_weight :751-774; sp_stage :830-856 -> warp :776-828) is called, followed by ``backward`` of
``sum(output * G)`` with fixed cotangents G.  The two deform NETWORKS are not part of this fixture (mlp.npz / sp_deformnet.npz pin
them): ``model.sk_deform_net`` / ``model.sp_deform_net`` are replaced by a module that returns stored tensors, so their outputs are
leaves whose gradients are recorded.

What is independent of the stand-ins in it (checked by the script itself, and again by tests/test_lietorch_standin.py):
  * skeleton_warp_SE3 == the reference's 4x4 twin ``skeleton_warp`` (sk_gs.py:182-190, pure torch) on the same chain, values AND
    the gradients w.r.t. the 7-vectors (through ``ops_3d.rigid.quaternion_to_Rt`` of the normalised quaternion);
  * the skinning == the reference's own matrix branch of ``warp`` (sk_gs.py:806-810: ``ops_3d.apply(points[:, None], spT[indices])``).
Per scenario the file holds ``<name>/in/*`` (every tensor needed to replay without the reference), ``<name>/out/*`` and ``<name>/grad/*``.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (stub finder only)

REF = make_golden.REF

NET_CFG = dict(pos_enc_p='freq_torch', pos_enc_p_cfg={'degree': 10}, pos_enc_t='freq_torch', pos_enc_t_cfg={'degree': 6}, width=32,
               depth=8, skips=[4])

# name -> (stage, constructor options).  K = num_knn as exps/default.yaml:20 / the SC-GS and SP-GS configs (3)
SCENARIOS = {
    'sk_W': ('sk', dict(LBS_method='W', warp_method='LBS', sep_rot=False, num_knn=5, num_superpoints=20)),
    'sk_lie': ('sk', dict(LBS_method='W', warp_method='LBS', sep_rot=False, num_knn=5, num_superpoints=20, which_rotation='lie')),
    'sk_kernel': ('sk', dict(LBS_method='weighted_kernel', warp_method='LBS', sep_rot=False, num_knn=3, num_superpoints=20)),
    'sp_W_LBS': ('sp', dict(LBS_method='W', warp_method='LBS', sep_rot=False, num_knn=5, num_superpoints=24)),           # exps/default.yaml
    'sp_wk_LBSc_sep': ('sp', dict(LBS_method='weighted_kernel', warp_method='LBS_c', sep_rot=True, num_knn=3, num_superpoints=24)),  # d_nerf_sc_gs.yaml
    'sp_W_largest': ('sp', dict(LBS_method='W', warp_method='largest', sep_rot=False, num_knn=3, num_superpoints=24)),   # d_nerf_sp_gs.yaml
    'sp_dist_LBS': ('sp', dict(LBS_method='dist', warp_method='LBS', sep_rot=False, num_knn=5, num_superpoints=24)),
    'sp_kernel_LBSc': ('sp', dict(LBS_method='kernel', warp_method='LBS_c', sep_rot=False, num_knn=4, num_superpoints=24)),
}
P = 300


def main():
    assert os.path.isdir(REF), 'the reference is only mounted in the build container'
    sys.dont_write_bytecode = True
    make_golden.STUBS = make_golden.STUBS - {'lietorch', 'pytorch3d'}
    sys.meta_path.insert(0, make_golden._Finder())
    sys.path[:0] = [ROOT, REF]
    import warnings
    warnings.filterwarnings('ignore')
    import torch
    import sk_gs_amd
    sk_gs_amd.install_as_lietorch()
    sk_gs_amd.install_as_pytorch3d()
    import networks.sk_gs as sk                       # the reference, unmodified
    from my_ext import ops_3d
    from lietorch import SE3
    assert sk.SE3 is sk_gs_amd.lietorch.SE3 and sk.knn_points is sk_gs_amd.pytorch3d_ops.knn_points
    assert sk.__file__.startswith(REF)
    f32 = lambda t: t.detach().numpy().astype(np.float32)  # noqa: E731
    rec = {}

    class Fixed(torch.nn.Module):
        """stands where a deform network stood: returns the stored leaves (sk_deform_net: a tuple, kinematic :1075; sp_deform_net: a dict,
        sp_stage :846-849)"""
        is_blender = True

        def __init__(self, out):
            super().__init__()
            self.out = out

        def forward(self, x, t, **kw):
            return self.out

    for name, (stage, opts) in SCENARIOS.items():
        g = torch.Generator().manual_seed(sum(map(ord, name)) * 7919)
        rn = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale  # noqa: E731
        M, K = opts['num_superpoints'], opts['num_knn']
        model = sk.SkeletonGaussianSplatting(sh_degree=3, net_cfg=NET_CFG, sk_deform_net_cfg=NET_CFG, hyper_dim=8, is_blender=True,
                                             train_schedule={'sp': 10, 'sk': 10}, **opts)
        model.train()
        par = lambda t: torch.nn.Parameter(t.clone())  # noqa: E731
        # the Gaussians (networks/gaussian_splatting.py:134-139)
        model._xyz, model._scaling = par(rn(P, 3)), par(rn(P, 3, scale=0.3) - 3.0)
        model._rotation, model._opacity = par(torch.nn.functional.normalize(rn(P, 4), dim=-1)), par(rn(P, 1, scale=1.5))
        model._features_dc, model._features_rest = par(rn(P, 1, 3)), par(rn(P, 15, 3, scale=0.1))
        model.hyper_feature = par(rn(P, 8, scale=0.3))
        leaves = {'_xyz': model._xyz, '_scaling': model._scaling, '_rotation': model._rotation, '_opacity': model._opacity}
        if model.sp_W is not None:
            model.sp_W = par(rn(P, M))
            leaves['sp_W'] = model.sp_W
        if model._sp_radius is not None:
            model._sp_radius = par(rn(M, scale=0.3) - 0.5)
            leaves['_sp_radius'] = model._sp_radius
        if model._sp_weight is not None:
            model._sp_weight = par(rn(M))
            leaves['_sp_weight'] = model._sp_weight
        t = torch.tensor([[0.37]])
        ins = {'t': t, 'M': torch.tensor(M), 'K': torch.tensor(K)}
        if stage == 'sk':
            # a random tree and its 2^l-ancestor table (what joint_discovery leaves in joint_parents, sk_gs.py:106-131)
            parents = torch.tensor([0] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, M)])
            depth = 0
            d, cur = parents.clone(), torch.arange(M)
            while (d != cur).any():
                cur, d = d, parents[d]
                depth += 1
            L = max(1, int(np.ceil(np.log2(max(depth, 1) + 1))))
            table = torch.zeros(M, L, dtype=torch.long)
            table[:, 0] = parents
            for lv in range(1, L):
                table[:, lv] = table[table[:, lv - 1], lv - 1]
            model.joint_parents, model.joint_root = table.int(), torch.tensor(0, dtype=torch.int32)
            model.sk_is_init = torch.tensor(True)
            model.joints = par(rn(M, 3, scale=0.7))
            frames, time_id = 3, 1
            gt = rn(frames, 7, scale=0.2)
            gt[:, 3:] = torch.nn.functional.normalize(gt[:, 3:] + torch.tensor([0, 0, 0, 1.]), dim=-1)
            model.global_tr = par(gt)
            model.sk_cache = torch.zeros(frames, M, sum(model.sk_dims))
            model.num_frames = frames
            Rdim = model.sk_dims[0]
            net_out = (rn(M, Rdim, scale=0.3).requires_grad_(), rn(M, 4, scale=0.05).requires_grad_(), rn(M, 3, scale=0.01).requires_grad_())
            model.sk_deform_net = Fixed(net_out)
            leaves.update(joints=model.joints, global_tr=model.global_tr, net_sk_r=net_out[0], net_d_rot=net_out[1], net_d_scale=net_out[2])
            ins.update(parents_table=table, root=torch.tensor(0), time_id=torch.tensor(time_id))
            outputs = model.forward(t=t, stage='sk', time_id=time_id)
            keys = ['points', 'scales', 'rotations', 'opacity', '_skT', '_knn_w', '_sk_rot', '_sk_scale', '_d_xyz', '_d_rot', '_d_scale']
        else:
            model.sp_points = par(rn(M, 3, scale=0.8))
            model.sp_hyper_feature = par(rn(M, 8, scale=0.3))
            net_out = {'d_xyz': rn(M, 3, scale=0.2).requires_grad_(), 'd_rotation': rn(M, 4, scale=0.3).requires_grad_(),
                       'd_scaling': rn(M, 3, scale=0.01).requires_grad_()}
            if opts['sep_rot']:
                net_out['g_rotation'] = rn(M, 4, scale=0.3).requires_grad_()
            model.sp_deform_net = Fixed(net_out)
            leaves.update(sp_points=model.sp_points, hyper_feature=model.hyper_feature, sp_hyper_feature=model.sp_hyper_feature,
                          **{'net_' + k: v for k, v in net_out.items()})
            outputs = model.forward(t=t, stage='sp', time_id=0)
            keys = ['points', 'scales', 'rotations', 'opacity', '_spT', '_knn_w', '_sp_scale']
            if opts['sep_rot']:
                keys.append('_sp_rot')
            if opts['warp_method'] == 'largest':
                rec[f'{name}/out/p2sp'] = model.p2sp.numpy().astype(np.int64)
        if name == 'sp_W_LBS':
            # the reference's Lie-group regulariser on the superpoint transforms (loss_sp_arap, sk_gs.py:1371-1381): inv, product, log
            # and act of the stand-in under the reference's own code -- values and the gradient that reaches the transforms
            spT_leaf = outputs['_spT'].detach().clone().requires_grad_()
            arap, arap_ct = model.loss_sp_arap(SE3.InitFromVec(spT_leaf))
            (g_spT,) = torch.autograd.grad(arap + 0.5 * arap_ct, spT_leaf)   # (not .backward(): the scenario's own gradients follow)
            rec['arap/spT'], rec['arap/sp_points'] = f32(spT_leaf), f32(model.sp_points)
            rec['arap/sk_knn_num'] = np.int64(model.sk_knn_num)
            rec['arap/loss'], rec['arap/loss_ct'], rec['arap/g_spT'] = f32(arap), f32(arap_ct), f32(g_spT)
            print(f'   loss_sp_arap {float(arap):.6f} {float(arap_ct):.6f}')
        for k, v in ins.items():
            rec[f'{name}/in/{k}'] = v.numpy()
        for k, v in leaves.items():
            rec[f'{name}/in/{k}'] = f32(v)
        rec[f'{name}/out/_knn_i'] = outputs['_knn_i'].numpy().astype(np.int64)
        loss = 0
        for k in keys:
            G = rn(*outputs[k].shape)
            rec[f'{name}/cot/{k}'] = f32(G)
            rec[f'{name}/out/{k}'] = f32(outputs[k])
            if outputs[k].requires_grad:
                loss = loss + (outputs[k] * G).sum()
        loss.backward()
        for k, v in leaves.items():
            if v.grad is not None:
                rec[f'{name}/grad/{k}'] = f32(v.grad)
        print(f'{name:16s} stage {stage}  loss {float(loss):+.5f}  grads: {sorted(k for k, v in leaves.items() if v.grad is not None)}')

        # ---- checks that do not involve the stand-ins' conventions --------------------------------------------------
        if stage == 'sk' and name == 'sk_W':
            # (1) skeleton_warp_SE3 against the reference's 4x4 twin, values and 7-vector gradients
            v = torch.cat([rn(M, 3, scale=0.5), torch.nn.functional.normalize(rn(M, 4), dim=-1)], dim=-1)
            gv = torch.cat([rn(3, scale=0.3), torch.nn.functional.normalize(rn(4), dim=-1)])
            a, ga = v.clone().requires_grad_(), gv.clone().requires_grad_()
            out_se3 = sk.skeleton_warp_SE3(SE3.InitFromVec(a), SE3.InitFromVec(ga), table, torch.tensor(0))
            Gm = rn(M, 3, 4)
            (out_se3.matrix()[:, :3, :] * Gm).sum().backward()
            b, gb = v.clone().requires_grad_(), gv.clone().requires_grad_()
            to_Rt = lambda x: ops_3d.rigid.quaternion_to_Rt(torch.cat([x[..., :3], torch.nn.functional.normalize(x[..., 3:], dim=-1)], -1))  # noqa: E731
            out_mat = sk.skeleton_warp(to_Rt(b), to_Rt(gb), table, torch.tensor(0))
            (out_mat[:, :3, :] * Gm).sum().backward()
            e_val = float((out_se3.matrix() - out_mat).abs().max())
            e_g = float((a.grad - b.grad).abs().max() / b.grad.abs().max())
            e_gg = float((ga.grad - gb.grad).abs().max() / gb.grad.abs().max())
            print(f'   skeleton_warp_SE3 vs skeleton_warp (4x4): value {e_val:.2e}, grad local {e_g:.2e}, grad global {e_gg:.2e}')
            assert e_val < 2e-6 and e_g < 2e-5 and e_gg < 2e-5
            for k, val in dict(local=v, glob=gv, parents_table=table, cot=Gm, out=out_mat[:, :3, :], g_local=b.grad, g_glob=gb.grad).items():
                rec[f'chain4x4/{k}'] = val.detach().numpy().astype(np.int64 if k == 'parents_table' else np.float32)
            # (2) the skinning against the matrix branch of warp (sk_gs.py:806-810): spT as a Tensor of 4x4s
            skT = outputs['_skT'].detach()
            idx, w = outputs['_knn_i'], outputs['_knn_w'].detach()
            pts = model._xyz.detach()
            d_mat = (ops_3d.apply(pts[:, None], to_Rt(skT)[idx]) * w[..., None]).sum(dim=1) - pts
            e = float((d_mat - outputs['_d_xyz']).abs().max())
            print(f'   skinning vs the 4x4 branch of warp: {e:.2e}')
            assert e < 5e-6
    path = os.path.join(HERE, 'sk_stage.npz')
    if '--check' in sys.argv:  # tests/test_lietorch_standin.py: the reference, run again, still gives the committed file
        have = np.load(path)
        assert sorted(have.files) == sorted(rec), 'sk_stage.npz holds other arrays than this run'
        worst = 0.0
        for k, v in rec.items():
            if v.dtype.kind in 'iu':
                assert np.array_equal(have[k], v), k
            else:
                worst = max(worst, float(np.abs(have[k] - v).max() / max(1.0, float(np.abs(v).max()))))
        assert worst < 1e-6, worst
        print(f'SK-STAGE-CHECK-OK {len(rec)} arrays, worst {worst:.1e}')
        return
    np.savez_compressed(path, **rec)
    print('wrote', path, len(rec), 'arrays')


if __name__ == '__main__':
    main()
