#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's pure-torch helpers.

Run in the build container only (needs /root/reference, read-only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The GPU box never sees the reference: only the small .npz / .json files written here travel (they are data --
inputs and the reference's outputs -- not source).  Third-party modules the reference imports but this image lacks
(lietorch, pytorch3d, diff_gaussian_rasterization, cv2, ...) are replaced by inert stubs; nothing stubbed is ever
called for a fixture.

What is pinned (reference file:line of the function that produced the expected values):
  surface.json    networks/renderer/gaussian_render.py: NamedTuple fields/defaults, call signatures     :20-48,54-67,191-340
  sh.npz          networks/encoders/sphere_harmonics.py:130-185 eval_sh (degree 0..3)
  quaternion.npz  my_ext/ops_3d/quaternion.py:162-172 toR, :26-52 mul, :101-104 xfm; rigid.py:110-130 quaternion_to_Rt
  cov2d.npz       networks/GS_utils.py:102-125 compute_cov2D (row-major variant; [1,1] lacks the +0.3 there)
  camera.npz      my_ext/ops_3d/coord_trans_opencv.py:87-119 look_at, :203-239 perspective; coord_trans_common.py:56-60
  skeleton.npz    networks/sk_gs.py:167-190 skeleton_warp_v0 / skeleton_warp on 4x4 matrices; xfm.py:60-79 apply
  ssim.npz        networks/losses/ssim.py:20-62 SSIM_Loss, networks/losses/image_loss.py:6-32 ImageLoss('l1'):
                  loss values and d(0.8 L1 + 0.2 SSIM)/d image
  mlp.npz         my_ext/blocks/mlp.py:43-85 MLP_with_skips (4 x 16, skip after layer 2, heads 4|4|3): parameters by their
                  state_dict names, outputs and parameter gradients for fixed cotangents
  densify.npz     networks/gaussian_splatting.py:515-655 GaussianSplatting.change_optimizer / densify_and_clone / prune /
                  reset_opacity / densify_and_split run on CPU with torch.optim.Adam(eps=1e-15): parameters, Adam moments
                  and statistics after every operation, with Adam steps (recorded gradients) in between
  lr_schedule.npz networks/gaussian_splatting.py:56-84 get_expon_lr_func at a grid of steps (three parameter sets)
"""
import importlib.abc
import importlib.machinery
import inspect
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
STUBS = {'imageio', 'cv2', 'plyfile', 'lietorch', 'pytorch3d', 'pykdtree', 'diff_gaussian_rasterization', 'dearpygui',
         'seaborn', 'torchmetrics', 'skimage', 'matplotlib', 'open3d', 'trimesh', 'lpips', 'kornia', 'tensorboard',
         'tensorboardX'}


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        m = _Stub(self.__name__ + '.' + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Stub('call')


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


def main():
    assert os.path.isdir(REF), 'the reference is only mounted in the build container'
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings('ignore')
    import torch
    from my_ext import ops_3d
    from networks import GS_utils
    from networks.encoders.sphere_harmonics import eval_sh
    from networks.losses.image_loss import ImageLoss
    from networks.losses.ssim import SSIM_Loss
    from networks.renderer import gaussian_render as gr
    import networks.sk_gs as sk

    g = torch.Generator().manual_seed(20240501)
    f32 = lambda t: t.detach().numpy().astype(np.float32)  # noqa: E731

    # ---- operator surface ------------------------------------------------------------------------------------
    surface = {
        'GaussianRasterizationSettings._fields': list(gr.GaussianRasterizationSettings._fields),
        'GaussianRasterizationSettings._field_defaults': dict(gr.GaussianRasterizationSettings._field_defaults),
        'RasterizeBuffer._fields': list(gr.RasterizeBuffer._fields),
        '_RasterizeGaussians.forward': str(inspect.signature(gr._RasterizeGaussians.forward.__wrapped__
                                                              if hasattr(gr._RasterizeGaussians.forward, '__wrapped__')
                                                              else gr._RasterizeGaussians.forward)),
        'rasterize_gaussians': str(inspect.signature(gr.rasterize_gaussians)),
        'GaussianRasterizer.forward': str(inspect.signature(gr.GaussianRasterizer.forward)),
        'GaussianRasterizer.markVisible': str(inspect.signature(gr.GaussianRasterizer.markVisible)),
        'render': str(inspect.signature(gr.render)),
        'topk_weights': str(inspect.signature(gr.topk_weights)),
    }
    json.dump(surface, open(os.path.join(HERE, 'surface.json'), 'w'), indent=1, sort_keys=True)

    # ---- SH ------------------------------------------------------------------------------------------------------
    N = 256
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    sh = torch.randn(N, 16, 3, generator=g)  # our layout [P, S, 3]; the reference helper wants [..., C, S]
    out = {f'deg{d}': f32(eval_sh(d, sh.permute(0, 2, 1), dirs)) for d in range(4)}
    np.savez(os.path.join(HERE, 'sh.npz'), dirs=f32(dirs), sh=f32(sh), **out)

    # ---- quaternions / rigid -----------------------------------------------------------------------------------------
    q = torch.nn.functional.normalize(torch.randn(128, 4, generator=g), dim=-1)
    q2 = torch.nn.functional.normalize(torch.randn(128, 4, generator=g), dim=-1)
    pts = torch.randn(128, 3, generator=g)
    t = torch.randn(128, 3, generator=g)
    np.savez(os.path.join(HERE, 'quaternion.npz'), q=f32(q), q2=f32(q2), pts=f32(pts), t=f32(t),
             toR=f32(ops_3d.quaternion.toR(q)), mul=f32(ops_3d.quaternion.mul(q, q2)),
             xfm=f32(ops_3d.quaternion.xfm(pts, q)), Rt=f32(ops_3d.rigid.quaternion_to_Rt(q, t)),
             apply=f32(ops_3d.apply(pts, ops_3d.rigid.quaternion_to_Rt(q, t))))

    # ---- cov2D (row-major variant) -------------------------------------------------------------------------------
    Np = 100
    points = torch.randn(Np, 3, generator=g, dtype=torch.float64)
    L = torch.randn(Np, 3, 3, generator=g, dtype=torch.float64) * 0.2
    S = L @ L.transpose(-1, -2)
    cov3D = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
    eye = torch.tensor([0.5, -0.3, -4.0], dtype=torch.float64)
    Tw2v = ops_3d.opencv.look_at(eye.float(), torch.zeros(3)).double()  # at=None is broken upstream (index 3)
    size = 400
    fov = np.deg2rad(60.)
    focal = 0.5 * size / np.tan(0.5 * fov)
    tanfov = np.tan(0.5 * fov)
    cov2d = GS_utils.compute_cov2D(points, cov3D, Tw2v, focal, focal, tanfov, tanfov)
    np.savez(os.path.join(HERE, 'cov2d.npz'), points=points.numpy(), cov3D=cov3D.numpy(), Tw2v=Tw2v.numpy(),
             focal=focal, tanfov=tanfov, cov2d=cov2d.numpy())

    # ---- cameras -------------------------------------------------------------------------------------------------
    eyes = torch.tensor([[0., 0., -4.], [3.0, 1.0, 2.5], [-2.0, -3.0, 1.0]])
    Tw2v = torch.stack([ops_3d.opencv.look_at(e, torch.zeros(3)) for e in eyes])
    fovx = 0.6911
    persp = {}
    for (w, h) in [(800, 800), (512, 384), (200, 136)]:
        fovy = float(ops_3d.fovx_to_fovy(fovx, w / h))
        persp[f'persp_{w}x{h}'] = f32(ops_3d.opencv.perspective(fovy, n=2., f=6., size=(w, h)))
        persp[f'fovy_{w}x{h}'] = np.float64(fovy)
    np.savez(os.path.join(HERE, 'camera.npz'), eyes=f32(eyes), Tw2v=f32(Tw2v), fovx=fovx, **persp)

    # ---- skeleton chain on 4x4 matrices ----------------------------------------------------------------------------
    sk_out = {}
    for M in (1, 20, 32):
        gen = torch.Generator().manual_seed(M)
        parents = torch.zeros(M, dtype=torch.long)
        for i in range(1, M):
            parents[i] = int(torch.randint(0, i, (1,), generator=gen))
        ql = torch.nn.functional.normalize(torch.randn(M, 4, generator=gen) * 0.3 + torch.tensor([0, 0, 0, 1.]), dim=-1)
        tl = torch.randn(M, 3, generator=gen) * 0.3
        local = ops_3d.rigid.quaternion_to_Rt(ql, tl)
        qg = torch.nn.functional.normalize(torch.randn(4, generator=gen), dim=-1)
        tg = torch.randn(3, generator=gen) * 0.2
        G = ops_3d.rigid.quaternion_to_Rt(qg, tg)
        # 2^l-th ancestor table as joint_discovery builds it (sp_gs_joint.cu:55-85)
        depth = torch.zeros(M, dtype=torch.long)
        for i in range(M):
            f, d = i, 0
            while f != 0:
                f = int(parents[f]); d += 1
            depth[i] = d
        Lv = 0
        while (1 << Lv) < int(depth.max()):
            Lv += 1
        Lv = max(Lv, 1)
        table = torch.zeros(M, Lv, dtype=torch.long)
        table[:, 0] = parents
        for l in range(1, Lv):
            table[:, l] = table[table[:, l - 1], l - 1]
        v0 = sk.skeleton_warp_v0(local, G, parents, torch.tensor(0))
        v1 = sk.skeleton_warp(local, G, table, torch.tensor(0))
        assert float((v0 - v1).abs().max()) < 1e-5
        sk_out.update({f'M{M}_parents': parents.numpy(), f'M{M}_table': table.numpy(), f'M{M}_ql': f32(ql),
                       f'M{M}_tl': f32(tl), f'M{M}_qg': f32(qg), f'M{M}_tg': f32(tg), f'M{M}_global': f32(v1)})
    np.savez(os.path.join(HERE, 'skeleton.npz'), **sk_out)

    # ---- image loss ------------------------------------------------------------------------------------------------
    ssim_f, l1_f = SSIM_Loss(), ImageLoss(method='l1')
    rec = {}
    for k in range(3):
        x = torch.rand(1, 64, 64, 3, generator=g, requires_grad=True)
        y = torch.rand(1, 64, 64, 3, generator=g)
        ls, ll = ssim_f(x, y), l1_f(x, y)
        total = 0.8 * ll + 0.2 * ls
        (grad,) = torch.autograd.grad(total, x)
        rec.update({f'x{k}': f32(x), f'y{k}': f32(y), f'ssim{k}': np.float64(ls.item()), f'l1{k}': np.float64(ll.item()),
                    f'total{k}': np.float64(total.item()), f'grad{k}': f32(grad)})
    np.savez(os.path.join(HERE, 'ssim.npz'), **rec)

    # ---- deform MLP (skeleton stage): MLP_with_skips as configured by SimpleDeformationNetwork -----------------------
    from my_ext.blocks.mlp import MLP_with_skips
    torch.manual_seed(20240502)
    net = MLP_with_skips(in_channels=12, dim_hidden=16, out_channels=(4, 4, 3), num_layers=4, skips=(2,))
    x = torch.randn(6, 12, generator=g, requires_grad=True)
    outs = net(x)
    gy = [torch.randn(o.shape, generator=g) for o in outs]
    grads = torch.autograd.grad(outs, list(net.parameters()), gy)
    rec = {'x': f32(x), **{f'out{j}': f32(o) for j, o in enumerate(outs)}, **{f'gy{j}': f32(v) for j, v in enumerate(gy)}}
    for (n, p), gr in zip(net.named_parameters(), grads):
        rec['param.' + n], rec['grad.' + n] = f32(p), f32(gr)
    np.savez(os.path.join(HERE, 'mlp.npz'), **rec)

    # ---- adaptive density control: the reference's own clone / prune / reset_opacity / split on a torch Adam ---------
    from networks.gaussian_splatting import GaussianSplatting
    gs = GaussianSplatting(sh_degree=3, use_official_gaussians_render=False)
    gd = torch.Generator().manual_seed(515)
    P0, names = 60, ['_xyz', '_features_dc', '_features_rest', '_opacity', '_scaling', '_rotation']
    shapes = {'_xyz': (3,), '_features_dc': (1, 3), '_features_rest': (15, 3), '_opacity': (1,), '_scaling': (3,),
              '_rotation': (4,)}
    init = {n: torch.randn(P0, *shapes[n], generator=gd) for n in names}
    init['_scaling'] = torch.log(torch.rand(P0, 3, generator=gd) * 0.09 + 0.005)  # max scale straddles 0.01 * extent
    init['_opacity'] = torch.randn(P0, 1, generator=gd) * 2.0
    for n in names:
        setattr(gs, n, torch.nn.Parameter(init[n].clone()))
    lrs = {'xyz': 0.16e-3, 'f_dc': 2.5e-3, 'f_rest': 2.5e-3 / 20, 'opacity': 50e-3, 'scaling': 5e-3, 'rotation': 1e-3}
    opt = torch.optim.Adam([{'params': [getattr(gs, n)], 'lr': lrs[gs.param_names_map[n]], 'name': gs.param_names_map[n]}
                            for n in names], eps=1e-15)
    rec = {'init.' + n: f32(init[n]) for n in names}
    rec['lr'] = np.array([lrs[gs.param_names_map[n]] for n in names], dtype=np.float64)

    def adam_step(tag):
        for n in names:
            p = getattr(gs, n)
            p.grad = torch.randn(p.shape, generator=gd)
            rec[f'{tag}.grad.{n}'] = f32(p.grad)
        opt.step()

    def snapshot(tag):
        for n in names:
            p = getattr(gs, n)
            st = opt.state[p]
            rec[f'{tag}.{n}'], rec[f'{tag}.m.{n}'], rec[f'{tag}.v.{n}'] = f32(p), f32(st['exp_avg']), f32(st['exp_avg_sq'])
        rec[f'{tag}.accum'], rec[f'{tag}.denom'] = f32(gs.xyz_gradient_accum), f32(gs.denom)
        rec[f'{tag}.radii'] = f32(gs.max_radii2D)

    def set_stats(tag):
        n = gs._xyz.shape[0]
        gs.xyz_gradient_accum = torch.rand(n, 1, generator=gd) * 4e-4
        gs.denom = torch.randint(0, 3, (n, 1), generator=gd).float()  # zeros -> nan -> 0, as in densify()
        gs.max_radii2D = torch.rand(n, generator=gd) * 30
        rec[f'{tag}.in_accum'], rec[f'{tag}.in_denom'] = f32(gs.xyz_gradient_accum), f32(gs.denom)
        rec[f'{tag}.in_radii'] = f32(gs.max_radii2D)

    extent, thr = 5.0, 2e-4
    with torch.no_grad():
        pass
    adam_step('s0'), adam_step('s1')
    snapshot('after_steps')
    set_stats('clone')
    with torch.no_grad():
        grads = gs.xyz_gradient_accum / gs.denom
        grads[grads.isnan()] = 0.0
        gs.densify_and_clone(opt, grads, thr, 0.01 * extent)
    snapshot('after_clone')
    set_stats('prune')
    with torch.no_grad():
        gs.prune(opt, min_opacity=0.05, extent=extent, max_screen_size=20.0)
    snapshot('after_prune')
    adam_step('s2')
    snapshot('after_step2')
    with torch.no_grad():
        gs.reset_opacity(opt)
    snapshot('after_reset')
    set_stats('split')
    torch.manual_seed(99)
    with torch.no_grad():
        grads = gs.xyz_gradient_accum / gs.denom
        grads[grads.isnan()] = 0.0
        n_before = gs._xyz.shape[0]
        big = (grads.squeeze() >= thr) & (torch.exp(gs._scaling).amax(1) > 0.01 * extent)
        rec['split.selected'] = big.numpy()
        gs.densify_and_split(opt, grads, thr, 0.01 * extent)
    snapshot('after_split')
    adam_step('s3')
    snapshot('after_step3')
    np.savez_compressed(os.path.join(HERE, 'densify.npz'), **rec)

    # ---- learning-rate schedule of the xyz group (get_expon_lr_func) --------------------------------------------------
    from networks.gaussian_splatting import get_expon_lr_func
    steps = np.array([-1, 0, 1, 10, 99, 100, 500, 2500, 15000, 29999, 30000, 40000], dtype=np.int64)
    cases = [dict(lr_init=0.16e-3, lr_final=0.0016e-3, lr_delay_mult=0.01, max_steps=30000),
             dict(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=200, lr_delay_mult=0.1, max_steps=5000),
             dict(lr_init=0.0, lr_final=0.0, max_steps=100)]
    rec = {'steps': steps}
    for i, c in enumerate(cases):
        f = get_expon_lr_func(**c)
        rec[f'lr{i}'] = np.array([f(int(t)) for t in steps], dtype=np.float64)
        rec[f'args{i}'] = np.array([c['lr_init'], c['lr_final'], c.get('lr_delay_steps', 0), c.get('lr_delay_mult', 1.0),
                                    c['max_steps']], dtype=np.float64)
    np.savez(os.path.join(HERE, 'lr_schedule.npz'), **rec)
    print('golden fixtures written to', HERE)
    for f in sorted(os.listdir(HERE)):
        print(f'  {f:<20} {os.path.getsize(os.path.join(HERE, f)):>8} B')


if __name__ == '__main__':
    main()
