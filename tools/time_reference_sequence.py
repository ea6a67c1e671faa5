#!/usr/bin/env python3
"""What the UNMODIFIED reference's deform costs on the MI355X through the stand-ins (sk_gs_amd.lietorch / pytorch3d_ops), at
config #1's size (100k Gaussians, 20 bones, K = 5) and at the superpoint stage's (512 superpoints, 3 + 8 search dimensions): the
call sequence of SkeletonGaussianSplatting.forward (benchlib/ref_sequence.py) forward + backward,

  * eager, as torch issues it (host-bound: the 20-row kinematic chain is ~150 small torch kernels forward, ~300 backward),
  * with the recognition of the skinning expression off (SKGS_LIE_FUSED=0 semantics: gather [P,K,7] + act + mul + sum as torch ops),
  * replayed as ONE hipGraph (torch.cuda.graph) -- what a user of the reference gets by wrapping the step in a graph,
  * and the fused operators of this package for the same arithmetic (sk_gs_amd.deform.lbs_deform + skeleton.bone_chain).
Usage: python tools/time_reference_sequence.py  (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402

from benchlib import ref_sequence as rs  # noqa: E402
from sk_gs_amd import lietorch as L, pytorch3d_ops as p3d, scene  # noqa: E402
from sk_gs_amd.deform import calc_lbs_weight, lbs_deform  # noqa: E402
from sk_gs_amd.skeleton import bone_chain, build_ancestor_table, build_topology  # noqa: E402

dev = torch.device('cuda')
torch.autograd.set_multithreading_enabled(False)


def timed(fn, n=50, warm=10):
    if fn is None:
        return float('nan'), float('nan')
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, 1e3 * t_host


def graphed(fn):
    """fn replayed as one hipGraph, or None when fn cannot be captured (the reference builds its constants with
    `x.new_tensor([...])`, a host-to-device copy inside the step: sk_gs.py:833,1076,196)"""
    try:
        return _graphed(fn)
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        print('   not capturable:', str(e).splitlines()[0])
        return None


def _graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def sk_inputs(P=100_000, M=20, seed=0):
    gs, bones = scene.make_gaussians(P, seed=seed), scene.make_bones(M, seed=seed)
    table, _ = build_ancestor_table(bones['parents'].long(), 0)
    leaf = lambda t: t.clone().to(dev).requires_grad_()  # noqa: E731
    g = torch.Generator().manual_seed(seed)
    a = {'_xyz': leaf(gs['xyz']), '_scaling': leaf(gs['log_scale']), '_rotation': leaf(gs['rot']), '_opacity': leaf(gs['opacity_logit']),
         'sp_W': leaf(torch.randn(P, M, generator=g)), 'joints': leaf(bones['joints']), 'net_sk_r': leaf(0.2 * torch.randn(M, 4, generator=g)),
         'net_d_rot': leaf(bones['d_rot']), 'net_d_scale': leaf(bones['d_scale']), 'global_tr': leaf(torch.tensor([[0.05, -0.1, 0.02, 0, 0, 0, 1.]])),
         'time_id': torch.tensor(0), 'parents_table': table.to(dev), 'root': torch.tensor(0)}
    cot = {k: torch.randn(P, n, generator=g).to(dev) for k, n in (('points', 3), ('scales', 3), ('rotations', 4), ('opacity', 1))}
    return a, cot, bones


def sp_inputs(P=100_000, M=512, seed=0):
    gs = scene.make_gaussians(P, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    leaf = lambda t: t.clone().to(dev).requires_grad_()  # noqa: E731
    a = {'_xyz': leaf(gs['xyz']), '_scaling': leaf(gs['log_scale']), '_rotation': leaf(gs['rot']), '_opacity': leaf(gs['opacity_logit']),
         'sp_points': leaf(gs['xyz'][torch.randperm(P, generator=g)[:M]]), 'hyper_feature': leaf(0.1 * torch.randn(P, 8, generator=g)),
         'sp_hyper_feature': leaf(0.1 * torch.randn(M, 8, generator=g)), '_sp_radius': leaf(torch.randn(M, generator=g) * 0.3 - 1.5),
         '_sp_weight': leaf(torch.randn(M, generator=g)), 'net_d_xyz': leaf(0.05 * torch.randn(M, 3, generator=g)),
         'net_d_rotation': leaf(0.1 * torch.randn(M, 4, generator=g)), 'net_d_scaling': leaf(0.01 * torch.randn(M, 3, generator=g))}
    cot = {k: torch.randn(P, n, generator=g).to(dev) for k, n in (('points', 3), ('scales', 3), ('rotations', 4), ('opacity', 1))}
    return a, cot


def sk_stage_accelerated(a):
    """rs.sk_stage with the reference's ``kinematic`` as sk_gs_amd.accelerate_reference() replaces it (reference_accel.kinematic on a
    stand-in `self`); the rest of the sequence -- search, weights, the skinning expression, the activations -- unchanged"""
    import types
    from sk_gs_amd import reference_accel as ra
    sys.modules.setdefault('lietorch', L)
    me = types.SimpleNamespace(training=True, test_time_interpolate=False, sk_feature=None, _R_dim=4, joint_parents=a['parents_table'].int(),
                               joint_root=a['root'], sk_cache=_cache(a), sk_deform_net=lambda x, t: (a['net_sk_r'], a['net_d_rot'], a['net_d_scale']))
    points = a['_xyz'].detach()
    sk_T, d_rot, d_scale = ra.kinematic(me, a['joints'], None, a['global_tr'][int(a['time_id'])].view(-1), int(a['time_id']), None)
    w, idx = rs._lbs_weights(p3d.knn_points, a, points, a['joints'], 5)
    d_xyz = (sk_T[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
    return rs._activate(a, d_xyz, (d_rot[idx] * w[..., None]).sum(dim=1), (d_scale[idx] * w[..., None]).sum(dim=1))


_CACHE = {}


def _cache(a):
    M = a['joints'].shape[0]
    if M not in _CACHE:
        _CACHE[M] = torch.zeros(1, M, 11, device=dev)
    return _CACHE[M]


def main():
    rows = []
    a, cot, bones = sk_inputs()
    params = [v for v in a.values() if torch.is_tensor(v) and v.requires_grad]

    def ref_step(stage_fn):
        for p in params:
            p.grad = None
        res = stage_fn()
        sum((res[k] * cot[k]).sum() for k in cot).backward()

    sk = lambda: rs.sk_stage(L, p3d.knn_points, a, 5)  # noqa: E731
    timed(lambda: ref_step(sk), n=30, warm=30)  # (a fresh process: the first timed region ran 1.5 x slower than every later one)
    rows.append(('sk 100k x 20: reference sequence on the stand-ins, eager', *timed(lambda: ref_step(sk))))
    rows.append(('sk 100k x 20: ... with accelerate_reference()\'s kinematic (the chain as one launch per direction)',
                 *timed(lambda: ref_step(lambda: sk_stage_accelerated(a)))))
    L._FUSED = False
    rows.append(('sk 100k x 20: ... skinning expression NOT recognised (generic torch ops)', *timed(lambda: ref_step(sk))))
    L._FUSED = True
    rows.append(('sk 100k x 20: ... as one hipGraph replay', *timed(graphed(lambda: ref_step(sk)))))
    topo = build_topology(bones['parents'].long(), 0, dev)

    def fused():
        for p in params:
            p.grad = None
        sk_T = bone_chain(a['net_sk_r'], a['joints'], a['global_tr'][0], topo)
        w, idx = calc_lbs_weight(a['_xyz'].detach(), a['joints'], 5, sp_W=a['sp_W'])
        out = lbs_deform(a['_xyz'].detach(), w, idx, sk_T, a['net_d_rot'], a['net_d_scale'], a['_xyz'], a['_scaling'], a['_rotation'], a['_opacity'])
        sum((o * cot[k]).sum() for o, k in zip(out, ('points', 'scales', 'rotations', 'opacity'))).backward()
    rows.append(('sk 100k x 20: this package\'s fused operators (bone_chain + calc_lbs_weight + lbs_deform), eager', *timed(fused)))
    rows.append(('sk 100k x 20: ... as one hipGraph replay', *timed(graphed(fused))))

    a, cot = sp_inputs()
    params = [v for v in a.values() if torch.is_tensor(v) and v.requires_grad]
    sp = lambda: rs.sp_stage(L, p3d.knn_points, a, 5, 'LBS', False)  # noqa: E731
    rows.append(('sp 100k x 512 (3+8 dims, weighted_kernel): reference sequence on the stand-ins, eager', *timed(lambda: ref_step(sp))))
    L._FUSED = False
    rows.append(('sp 100k x 512: ... skinning expression NOT recognised', *timed(lambda: ref_step(sp))))
    L._FUSED = True
    rows.append(('sp 100k x 512: ... as one hipGraph replay', *timed(graphed(lambda: ref_step(sp)))))
    # ---- the two deform networks: the modules' own torch forward against accelerate_reference()'s fast paths (modules with the
    # reference classes' structure: benchlib/ref_sequence.py; the kernels run on the modules' own parameters)
    from sk_gs_amd import reference_accel as ra
    from sk_gs_amd.superpoint import SpDeformNet
    g = torch.Generator().manual_seed(3)
    net = rs.RefSimpleDeformationNetwork().to(dev)
    pts, tt = (torch.rand(20, 3, generator=g) * 2 - 1).to(dev).requires_grad_(), torch.tensor([0.3125], device=dev)
    cots = [torch.randn(20, n, generator=g).to(dev) for n in (4, 4, 3)]
    ra._originals['sk_net'] = rs.RefSimpleDeformationNetwork.forward

    def net_step(fwd):
        for p_ in net.parameters():
            p_.grad = None
        pts.grad = None
        sum((o * c).sum() for o, c in zip(fwd(), cots)).backward()
    rows.append(('sk_deform_net, 20 joint rows, forward + backward: the module\'s own torch forward', *timed(lambda: net_step(lambda: net(pts, tt)))))
    rows.append(('sk_deform_net: ... through accelerate_reference() (one launch per direction)',
                 *timed(lambda: net_step(lambda: ra.simple_deform_forward(net, pts, tt)))))
    sp_net = SpDeformNet()
    sp_net.pos_enc_p, sp_net.pos_enc_t, sp_net.max_d_scale = rs.RefFreqEncoder(3, 10), rs.RefFreqEncoder(1, 6), -1.0
    sp_net = sp_net.to(dev)
    ra._originals['sp_net'] = lambda self, x, t, **kw: self.reference_forward(x, t)
    x512 = (torch.rand(512, 3, generator=g) * 2 - 1).to(dev)
    cot = {k: torch.randn(512, n, generator=g).to(dev) for k, n in (('d_xyz', 3), ('d_rotation', 4), ('d_scaling', 3))}

    def sp_net_step(fwd):
        for p_ in sp_net.parameters():
            p_.grad = None
        out = fwd()
        sum((out[k] * cot[k]).sum() for k in cot).backward()
    rows.append(('sp_deform_net, 512 superpoints, forward + backward: the module\'s own torch forward',
                 *timed(lambda: sp_net_step(lambda: sp_net.reference_forward(x512, tt)))))
    rows.append(('sp_deform_net: ... through accelerate_reference() (MFMA row blocks)',
                 *timed(lambda: sp_net_step(lambda: ra.deform_network_forward(sp_net, x512, tt)))))
    print(f'{"deform forward + backward":92s} {"GPU ms":>8s} {"host ms":>8s}')
    for name, gpu, host in rows:
        print(f'{name:92s} {gpu:8.3f} {host:8.3f}')
    print('fused_calls', L.fused_calls, 'hip_calls', p3d.hip_calls)


if __name__ == '__main__':
    main()
