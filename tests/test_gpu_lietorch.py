"""GPU tests of the lietorch / pytorch3d stand-ins (VERDICT r4 row b''): the reference's deform call sequence -- restated in
benchlib/ref_sequence.py, pinned on CPU against the reference's own run (tests/golden/sk_stage.npz) -- driven on the MI355X through
``sk_gs_amd.lietorch`` / ``sk_gs_amd.pytorch3d_ops``, where the skinning expression and the search are launches of libskgs_hip.so
(``skgs_se3_blend_forward/backward``, ``skgs_knn_bones``, ``skgs_sp_lbs_weights_forward``).

Tolerances: neighbour indices bit-exact; forward values <= 2e-6 max-norm relative against the fixture and the oracle (same
operation order); gradients <= 2e-5 (the bone gradient rows are summed by LDS atomics in an unspecified order); the north star's
bound is 1e-4.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from benchlib import ref_sequence as rs
from helpers import rel_err, to_np

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _mods():
    from sk_gs_amd import lietorch as L
    from sk_gs_amd import pytorch3d_ops as p3d
    return L, p3d


@pytest.mark.parametrize('name', sorted(rs.SCENARIOS))
def test_reference_sequence_on_the_gpu_matches_the_reference_run(name):
    L, p3d = _mods()
    z = np.load(os.path.join(GOLDEN, 'sk_stage.npz'))
    f0, h0 = dict(L.fused_calls), dict(p3d.hip_calls)
    res, out, got_grad, grad = rs.run_scenario(L, p3d.knn_points, z, name, device='cuda')
    assert L.fused_calls['forward'] == f0['forward'] + 1 and L.fused_calls['backward'] == f0['backward'] + 1   # the HIP launches ran
    assert L.fused_calls['materialised'] == f0['materialised']
    assert p3d.hip_calls['knn_bones'] + p3d.hip_calls['sp_search'] == h0['knn_bones'] + h0['sp_search'] + 1
    assert torch.equal(res['_knn_i'].cpu(), out['_knn_i'])
    for k, want in out.items():
        if k != '_knn_i':
            got = res[k].detach().cpu()
            assert rel_err(got.float() if got.is_floating_point() else got, want) <= 2e-6, (name, k)
    for k, want in grad.items():
        assert got_grad[k] is not None and rel_err(got_grad[k].cpu(), want) <= 2e-5, (name, k, rel_err(got_grad[k].cpu(), want))


@pytest.mark.parametrize('P,M,K,weighted,need_p', [(100_000, 20, 5, True, False), (100_000, 512, 5, True, True), (30_000, 1, 1, False, False),
                                                   (50_000, 1500, 3, True, True), (0, 7, 2, True, False), (4097, 1024, 16, True, False)])
def test_se3_blend_launch_against_the_generic_ops(oracle32, P, M, K, weighted, need_p):
    """skgs_se3_blend_forward / _backward (LDS path, the > 1024-bone path, no weights, an empty cloud) against the pure-torch group ops
    on the same device, and against the oracle's skinning"""
    L, _ = _mods()
    g = torch.Generator().manual_seed(P + M)
    v = torch.cat([0.4 * torch.randn(M, 3, generator=g), torch.randn(M, 4, generator=g)], -1)   # un-normalised q: the constructor's job
    idx = torch.randint(0, M, (P, K), generator=g)
    pts, w0, c = torch.randn(P, 3, generator=g), torch.softmax(torch.randn(P, K, generator=g), -1), torch.randn(P, 3, generator=g)

    def run(fused):
        a = v.cuda().requires_grad_()
        p = pts.cuda().requires_grad_(need_p)
        w = w0.cuda().requires_grad_() if weighted else None
        if fused:
            d = L._se3_blend(a, idx.cuda(), p, w)
        else:
            d = L._blend_reference(a, idx.cuda(), p, w)
        (d * c.cuda()).sum().backward()
        return d.detach(), a.grad, (w.grad if weighted else None), (p.grad if need_p else None)

    d1, ga1, gw1, gp1 = run(True)
    d0, ga0, gw0, gp0 = run(False)
    assert d1.shape == (P, 3) and ga1.shape == (M, 7)
    if P == 0:
        assert ga1.abs().sum() == 0
        return
    assert rel_err(d1, d0) <= 2e-6
    assert ga1[:, 6].abs().max() == 0                                   # tangent rows: slot 6 is empty
    assert rel_err(ga1, ga0) <= 2e-5, rel_err(ga1, ga0)
    if weighted:
        assert rel_err(gw1, gw0) <= 2e-6
    if need_p:
        assert rel_err(gp1, gp0) <= 2e-6
    if weighted and P <= 100_000:                                          # the oracle's skinning (d_xyz + p = the blend)
        n = min(P, 20_000)
        zeros = np.zeros((n, 3), np.float32)
        ref = oracle32.lbs_deform_forward(to_np(pts[:n]), to_np(w0[:n]), to_np(idx[:n]), to_np(v), np.zeros((M, 4), np.float32), np.zeros((M, 3), np.float32),
                                          to_np(pts[:n]), zeros, np.tile(np.float32([0, 0, 0, 1]), (n, 1)), np.zeros((n, 1), np.float32))
        assert rel_err(d1[:n].cpu() - pts[:n], ref['d_xyz']) <= 5e-6


@pytest.mark.parametrize('P,M,D,K', [(100_000, 20, 3, 5), (100_000, 512, 11, 5), (50_000, 512, 3, 3), (20_000, 64, 11, 8), (3000, 2000, 3, 4)])
def test_knn_points_on_the_gpu(oracle32, P, M, D, K):
    """pytorch3d.ops.knn_points's per-frame call shape: both HIP searches against the oracle (indices bit-exact), gradients through the
    distances against the closed form"""
    _, p3d = _mods()
    g = torch.Generator().manual_seed(P + M + D)
    p1, p2 = torch.randn(P, D, generator=g), torch.randn(M, D, generator=g)
    before = dict(p3d.hip_calls)
    a, b = p1.cuda().requires_grad_(), p2.cuda().requires_grad_()
    r = p3d.knn_points(a[None], b[None], None, None, K=K)
    took = 'sp_search' if (60 < M <= 1024 and D in (3, 11)) else 'knn_bones'
    assert p3d.hip_calls[took] == before[took] + 1
    d_ref, i_ref = oracle32.knn_bones(to_np(p1), to_np(p2), K)
    assert r.idx.dtype == torch.int64 and r.idx.shape == (1, P, K) and r.knn is None
    np.testing.assert_array_equal(to_np(r.idx[0]), i_ref)
    assert rel_err(r.dists[0], d_ref) <= 2e-6
    c = torch.randn(P, K, generator=g)
    (r.dists[0] * c.cuda()).sum().backward()
    diff = p1[:, None, :] - p2[torch.from_numpy(i_ref)]
    assert rel_err(a.grad, (2 * c[..., None] * diff).sum(1)) <= 2e-6
    gb = torch.zeros_like(p2).index_add_(0, torch.from_numpy(i_ref).reshape(-1), (-2 * c[..., None] * diff).reshape(-1, D))
    assert rel_err(b.grad, gb) <= 2e-5
    nn = p3d.knn_points(a[None], b[None], K=K, return_nn=True).knn
    assert torch.equal(nn[0].cpu(), p2[torch.from_numpy(i_ref)])


def test_reference_sequence_equals_the_fused_operators():
    """the stand-in route (what the unmodified reference executes) and this package's fused operators (sk_gs_amd.deform.lbs_deform +
    skeleton.bone_chain, what bench.py times) give the same deformed Gaussians and the same parameter gradients at config #1's size"""
    L, p3d = _mods()
    from sk_gs_amd import scene
    from sk_gs_amd.deform import calc_lbs_weight, lbs_deform
    from sk_gs_amd.skeleton import bone_chain, build_ancestor_table, build_topology
    P, M, K = 100_000, 20, 5
    gs, bones = scene.make_gaussians(P, seed=11), scene.make_bones(M, seed=11)
    g = torch.Generator().manual_seed(11)
    parents = bones['parents'] if 'parents' in bones else torch.tensor([0] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, M)])
    table, _ = build_ancestor_table(parents.long(), 0)
    dev = torch.device('cuda')
    leaf = lambda t: t.clone().to(dev).requires_grad_()  # noqa: E731
    mk = lambda: {'_xyz': leaf(gs['xyz']), '_scaling': leaf(gs['log_scale']), '_rotation': leaf(gs['rot']), '_opacity': leaf(gs['opacity_logit']),  # noqa: E731
                  'sp_W': leaf(torch.randn(P, M, generator=torch.Generator().manual_seed(12))), 'joints': leaf(bones['joints']),
                  'net_sk_r': leaf(0.2 * torch.randn(M, 4, generator=torch.Generator().manual_seed(13))), 'net_d_rot': leaf(bones['d_rot']),
                  'net_d_scale': leaf(bones['d_scale']),
                  'global_tr': leaf(torch.tensor([[0.05, -0.1, 0.02, 0.0, 0.0, 0.0, 1.0]])), 'time_id': torch.tensor(0),
                  'parents_table': table.to(dev), 'root': torch.tensor(0)}
    cot = {k: torch.randn(P, n, generator=g).to(dev) for k, n in (('points', 3), ('scales', 3), ('rotations', 4), ('opacity', 1))}
    a = mk()
    res = rs.sk_stage(L, p3d.knn_points, a, K)
    sum((res[k] * cot[k]).sum() for k in cot).backward()
    b = mk()
    topo = build_topology(parents.long(), 0, dev)
    sk_T = bone_chain(b['net_sk_r'], b['joints'], b['global_tr'][0], topo)
    w, idx = calc_lbs_weight(b['_xyz'].detach(), b['joints'], K, sp_W=b['sp_W'])
    means, scales, rotations, opacity = lbs_deform(b['_xyz'].detach(), w, idx, sk_T, b['net_d_rot'], b['net_d_scale'], b['_xyz'], b['_scaling'],
                                                   b['_rotation'], b['_opacity'])
    ((means * cot['points']).sum() + (scales * cot['scales']).sum() + (rotations * cot['rotations']).sum() + (opacity * cot['opacity']).sum()).backward()
    assert torch.equal(idx, res['_knn_i'])
    for got, k in ((means, 'points'), (scales, 'scales'), (rotations, 'rotations'), (opacity, 'opacity')):
        assert rel_err(res[k], got) <= 2e-6, k
    for k in ('_xyz', '_scaling', '_rotation', '_opacity', 'sp_W', 'net_d_rot', 'net_d_scale', 'net_sk_r', 'joints', 'global_tr'):
        assert rel_err(a[k].grad, b[k].grad) <= 5e-5, (k, rel_err(a[k].grad, b[k].grad))


@pytest.mark.parametrize('group', ['SO3', 'SE3'])
def test_hip_group_operators_against_the_torch_bodies(group):
    """csrc/lie_ops.hip (one launch per operator and direction) against the pure-torch bodies evaluated in fp64 on the CPU: every
    operator of both groups, values and gradients (tangent rows for group arguments), incl. the small-angle series"""
    L, _ = _mods()
    G = {'SO3': L.SO3, 'SE3': L.SE3}[group]
    K, N = G._math.K, G._math.N
    g = torch.Generator().manual_seed(17)
    B = 3000
    a = torch.randn(B, K, generator=g, dtype=torch.float64) * 0.8
    a[:40, -3:] *= 1e-8                                                   # theta < 1e-6: the series branches
    X = G.exp(a).data
    Y = G.exp(torch.randn(B, K, generator=g, dtype=torch.float64)).data
    X[:, -4:] *= 1.0 + 0.3 * torch.rand(B, 1, generator=g, dtype=torch.float64)   # un-normalised on purpose: the constructors normalise
    tang = torch.randn(B, K, generator=g, dtype=torch.float64)
    p3, p4 = torch.randn(B, 3, generator=g, dtype=torch.float64), torch.randn(B, 4, generator=g, dtype=torch.float64)
    cases = {
        'exp': (lambda x, y: G.exp(x).data, a, None),
        'log': (lambda x, y: G(x).log(), X, None),
        'inv': (lambda x, y: G(x).inv().data, X, None),
        'mul': (lambda x, y: (G(x) * G(y)).data, X, Y),
        'adj': (lambda x, y: G(x).adj(y), X, tang),
        'adjT': (lambda x, y: G(x).adjT(y), X, tang),
        'act': (lambda x, y: G(x).act(y), X, p3),
        'act4': (lambda x, y: G(x).act(y), X, p4),
        'vec': (lambda x, y: G(x).vec(), X, None),
        'InitFromVec.act': (lambda x, y: G.InitFromVec(x).act(y), X, p3),
    }
    for name, (f, x0, y0) in cases.items():
        res = {}
        for where in ('hip', 'ref'):
            conv = (lambda t: t.float().cuda()) if where == 'hip' else (lambda t: t.clone())
            x = conv(x0).requires_grad_()
            y = None if y0 is None else conv(y0).requires_grad_()
            before = dict(L.hip_op_calls)
            out = f(x, y)
            c = torch.randn(out.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
            (out * conv(c)).sum().backward()
            if where == 'hip':
                assert L.hip_op_calls['forward'] > before['forward'] or name == 'vec', name
                assert L.hip_op_calls['backward'] > before['backward'], name
            res[where] = (out.detach().cpu().double(), x.grad.cpu().double(), None if y is None else y.grad.cpu().double())
        (o1, gx1, gy1), (o0, gx0, gy0) = res['hip'], res['ref']
        if name in ('exp', 'inv', 'mul'):                                  # q and -q are the same rotation
            s = torch.sign((o1[:, -4:] * o0[:, -4:]).sum(-1, keepdim=True))
            o1 = torch.cat([o1[:, :-4], o1[:, -4:] * s], -1)
        if name != 'vec':
            assert rel_err(o1, o0) <= 3e-6, (group, name, rel_err(o1, o0))
        assert rel_err(gx1, gx0) <= 2e-5, (group, name, 'dX', rel_err(gx1, gx0))
        if gy0 is not None:
            assert rel_err(gy1, gy0) <= 2e-5, (group, name, 'dY', rel_err(gy1, gy0))


@pytest.mark.parametrize('R,C,M', [(500_000, 4, 20), (500_000, 1, 512), (300_000, 3, 1), (1000, 14, 5000), (0, 4, 7), (100_000, 40, 512)])
def test_typed_neighbour_indices_give_the_same_gather_gradients(R, C, M):
    """knn_points' typed indices: `table[indices]` forward is torch's gather, backward skgs_index_add_rows (LDS partials; global atomics
    beyond 48 KB of rows) -- against torch's own index backward"""
    _, p3d = _mods()
    g = torch.Generator().manual_seed(R + C + M)
    idx = torch.randint(0, M, (max(R // 5, 0), 5) if R % 5 == 0 else (R, 1), generator=g).cuda()
    table0 = torch.randn((M, C) if C > 1 else (M,), generator=g)
    cot = torch.randn(tuple(idx.shape) + ((C,) if C > 1 else ()), generator=g).cuda()
    res = []
    for typed in (True, False):
        t = table0.cuda().requires_grad_()
        i = p3d.NeighbourIndex.wrap(idx) if typed else idx
        before = p3d.hip_calls['gather_backward']
        out = t[i]
        assert type(out) is torch.Tensor
        (out * cot).sum().backward()
        assert p3d.hip_calls['gather_backward'] == before + int(typed)
        res.append((out.detach(), t.grad))
    assert torch.equal(res[0][0], res[1][0])
    assert rel_err(res[0][1], res[1][1]) <= 2e-5, rel_err(res[0][1], res[1][1])
    i = p3d.NeighbourIndex.wrap(idx)
    assert type(i.detach()) is torch.Tensor and type(i + 1) is torch.Tensor and (R == 0 or type(i[0]) is p3d.NeighbourIndex)
    assert type(table0.cuda()[i]) is torch.Tensor                       # no gradient asked for: torch's own gather


def test_reference_arap_loss_on_the_gpu():
    """loss_sp_arap (sk_gs.py:1371-1381; inv, product, log, act through csrc/lie_ops.hip) against the reference's own CPU run"""
    L, _ = _mods()
    z = np.load(os.path.join(GOLDEN, 'sk_stage.npz'))
    before = dict(L.hip_op_calls)
    spT = torch.from_numpy(z['arap/spT']).cuda().requires_grad_()
    loss, loss_ct = rs.loss_sp_arap(L, spT, torch.from_numpy(z['arap/sp_points']).cuda(), int(z['arap/sk_knn_num']))
    (loss + 0.5 * loss_ct).backward()
    assert L.hip_op_calls['forward'] >= before['forward'] + 4 and L.hip_op_calls['backward'] >= before['backward'] + 4
    assert abs(float(loss) - float(z['arap/loss'])) < 5e-6 and abs(float(loss_ct) - float(z['arap/loss_ct'])) < 5e-6
    assert rel_err(spT.grad.cpu(), z['arap/g_spT']) <= 5e-5


def test_accelerated_kinematic_equals_the_reference_chain():
    """sk_gs_amd.accelerate_reference()'s replacement of SkeletonGaussianSplatting.kinematic (one bone-chain launch per direction,
    handed back as SE3.InitFromVec) on a stand-in `self` with the reference run's inputs (sk_stage.npz, scenario sk_W): the bone
    transforms, the cache row and -- through the rest of the reference's sequence -- every gradient of the reference's own run"""
    import sys
    import types
    L, p3d = _mods()
    from sk_gs_amd import reference_accel as ra
    z = np.load(os.path.join(GOLDEN, 'sk_stage.npz'))
    a, out, cot, grad = rs.load_scenario(z, 'sk_W', 'cuda')
    saved = sys.modules.get('lietorch')
    sys.modules['lietorch'] = L
    try:
        me = types.SimpleNamespace(training=True, test_time_interpolate=False, sk_feature=None, _R_dim=4, joint_parents=a['parents_table'].int(),
                                   joint_root=a['root'].int().cuda(), sk_cache=torch.zeros(3, 20, 11, device='cuda'),
                                   sk_deform_net=lambda x, t: (a['net_sk_r'], a['net_d_rot'], a['net_d_scale']))
        tid = int(a['time_id'])
        before = dict(ra.calls)
        sk_T, d_rot, d_scale = ra.kinematic(me, a['joints'], a['t'].cuda(), a['global_tr'][tid].view(-1), tid, None)
        assert ra.calls['kinematic_fused'] == before['kinematic_fused'] + 1 and isinstance(sk_T, L.SE3)
        got = sk_T.vec()
        sign = torch.sign((got[:, 3:] * out['_skT'].cuda()[:, 3:]).sum(-1, keepdim=True))
        assert rel_err(got[:, :3], out['_skT'][:, :3]) <= 2e-6 and rel_err(got[:, 3:] * sign, out['_skT'][:, 3:]) <= 2e-6
        want_cache = torch.cat([F.normalize(a['net_sk_r'].detach() + torch.tensor([0, 0, 0, 1.], device='cuda'), dim=-1), a['net_d_rot'].detach(),
                                a['net_d_scale'].detach()], -1)
        assert torch.equal(me.sk_cache[tid], want_cache) and float(me.sk_cache[0].abs().max()) == 0
        # the rest of sk_stage on top of it (the reference's own lines, restated): the skinning and the activations
        points = a['_xyz'].detach()
        w, idx = rs._lbs_weights(p3d.knn_points, a, points, a['joints'], 5)
        d_xyz = (sk_T[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
        res = rs._activate(a, d_xyz, (d_rot[idx] * w[..., None]).sum(dim=1), (d_scale[idx] * w[..., None]).sum(dim=1))
        res.update(_skT=sk_T.vec(), _knn_w=w, _sk_rot=d_rot, _sk_scale=d_scale, _d_xyz=d_xyz, _d_rot=(d_rot[idx] * w[..., None]).sum(dim=1),
                   _d_scale=(d_scale[idx] * w[..., None]).sum(dim=1))
        loss = sum((res[k] * G.cuda()).sum() for k, G in cot.items() if k in res and res[k].requires_grad and k != '_skT')
        loss = loss + (res['_skT'] * (cot['_skT'].cuda() * torch.cat([torch.ones_like(sign).expand(-1, 3), sign.expand(-1, 4)], 1))).sum()
        loss.backward()
        for k, want in grad.items():
            assert a[k].grad is not None and rel_err(a[k].grad.cpu(), want) <= 3e-5, (k, rel_err(a[k].grad.cpu(), want))
        # outside the fast path's conditions the reference's own method is called
        ra._originals['kinematic'] = lambda *args: 'reference'
        assert ra.kinematic(me, a['joints'], a['t'].cuda(), None, tid, torch.zeros(20, 3, device='cuda')) == 'reference'
    finally:
        ra._originals.pop('kinematic', None)
        if saved is None:
            sys.modules.pop('lietorch', None)
        else:
            sys.modules['lietorch'] = saved


def test_accelerated_ssim_loss_equals_the_reference_loss():
    """accelerate_reference()'s SSIM_Loss.forward: one HWC image pair as sk_gs.loss hands it over (sk_gs.py:1527-1529) through the
    fused kernels, against the values and gradients the reference's own SSIM_Loss produced (tests/golden/ssim.npz)"""
    import types
    from sk_gs_amd import reference_accel as ra
    g = np.load(os.path.join(GOLDEN, 'ssim.npz'))
    me = types.SimpleNamespace(window_size=11, reduction='mean')
    from sk_gs_amd.losses import ssim_loss as ssim_loss_torch      # the torch restatement of ssim.py:20-62 (pinned by the same fixture)
    k = 0
    while f'x{k}' in g.files:
        x, y = torch.from_numpy(g[f'x{k}']).cuda(), torch.from_numpy(g[f'y{k}']).cuda()          # [1,H,W,3] as sk_gs.loss hands them over
        xh = x.clone().requires_grad_()
        before = ra.calls['ssim_fused']
        loss = ra.ssim_loss_forward(me, xh, y)
        assert ra.calls['ssim_fused'] == before + 1
        assert abs(float(loss) - float(g[f'ssim{k}'])) <= 2e-6, (float(loss), float(g[f'ssim{k}']))
        loss.backward()
        xt = x.clone().requires_grad_()
        ssim_loss_torch(xt[0].permute(2, 0, 1), y[0].permute(2, 0, 1)).backward()
        assert rel_err(xh.grad, xt.grad) <= 2e-5
        k += 1
    assert k >= 1
    ra._originals['ssim'] = lambda self, a, b: 'reference'
    try:
        assert ra.ssim_loss_forward(me, torch.rand(1, 8, 8, 3), torch.rand(1, 8, 8, 3)) == 'reference'      # CPU tensors
        assert ra.ssim_loss_forward(types.SimpleNamespace(window_size=7, reduction='mean'), x, y) == 'reference'
    finally:
        ra._originals.pop('ssim', None)


_RefFreqEncoder, _RefSimpleDeformationNetwork = rs.RefFreqEncoder, rs.RefSimpleDeformationNetwork


def test_accelerated_deform_networks_run_on_the_reference_modules_parameters():
    """accelerate_reference()'s two network fast paths on modules with the reference classes' structure: the one-launch 20-row kernels
    under ``SimpleDeformationNetwork.forward`` and the MFMA row-block kernels under ``DeformNetwork.forward`` give the modules' own
    torch forward (values, every parameter gradient, the joints' gradient through the input) -- on the modules' OWN parameter objects:
    an in-place update of a weight is what the next call computes with"""
    from sk_gs_amd import reference_accel as ra
    from sk_gs_amd.superpoint import SpDeformNet
    g = torch.Generator().manual_seed(4)
    before = dict(ra.calls)
    try:
        # ---- stage sk: 20 joints
        torch.manual_seed(1)
        net = _RefSimpleDeformationNetwork().cuda()
        with torch.no_grad():
            for h in net.dynamic_net.last:
                h.weight.normal_(0, 0.05, generator=None)
        ra._originals['sk_net'] = _RefSimpleDeformationNetwork.forward
        pts = (torch.rand(20, 3, generator=g) * 2 - 1).cuda().requires_grad_()
        t = torch.tensor([0.3125], device='cuda')
        cots = [torch.randn(20, n, generator=g).cuda() for n in (4, 4, 3)]
        for rep in range(2):
            got = ra.simple_deform_forward(net, pts, t)
            want = net(pts, t)
            assert len(got) == 3 and all(rel_err(a, b) <= 2e-5 for a, b in zip(got, want))
            params = list(net.parameters())
            gg = torch.autograd.grad(sum((a * c).sum() for a, c in zip(got, cots)), params + [pts])
            gw = torch.autograd.grad(sum((a * c).sum() for a, c in zip(want, cots)), params + [pts])
            for (n, _), a, b in zip(list(net.named_parameters()) + [('points', None)], gg, gw):
                assert rel_err(a, b) <= 1e-4, (rep, n, rel_err(a, b))
            with torch.no_grad():   # an optimizer step in place: hidden layer, head, bias
                net.dynamic_net.net[3].weight.add_(0.01 * torch.randn(256, 256, generator=g).cuda())
                net.dynamic_net.last[1].weight.mul_(1.5)
                net.dynamic_net.last[2].bias.add_(0.2)
        assert ra.calls['sk_net_fused'] == before['sk_net_fused'] + 2
        # outside the fast path: the module's own forward, counted
        n_ref = ra.calls['sk_net_reference']
        out2 = ra.simple_deform_forward(net, pts.detach()[None].expand(1, 20, 3).reshape(1, 20, 3), t)    # 3-d points: not the kernels' call
        assert ra.calls['sk_net_reference'] == n_ref + 1 and out2[0].shape == (1, 20, 4)
        # ---- stage sp: 512 superpoints, both network variants (gradients compared on draws in which the two fp32 evaluations took the
        # same side of every ReLU: helpers.sp_net_relu_masks_agree)
        from helpers import sp_net_relu_masks_agree
        compared = 0
        for blender, sep, tdeg in ((True, False, 6), (False, True, 10)):
            for seed in range(4):
                torch.manual_seed(2 + 100 * seed)
                ref = SpDeformNet(t_degree=tdeg, sep_rot=sep, is_blender=blender)
                ref.pos_enc_p, ref.pos_enc_t, ref.max_d_scale = _RefFreqEncoder(3, 10), _RefFreqEncoder(1, tdeg), -1.0
                with torch.no_grad():
                    for h in (ref.gaussian_warp, ref.gaussian_rotation, ref.gaussian_scaling) + ((ref.local_rotation,) if sep else ()):
                        h.weight.normal_(0, 0.05)
                    for layer in ref.linear:
                        layer.bias.normal_(0, 0.05)
                ref = ref.cuda()
                ra._originals['sp_net'] = lambda self, x, t, **kw: self.reference_forward(x, t)
                x = (torch.rand(512, 3, generator=g) * 2 - 1).cuda()
                keys = ('d_xyz', 'd_rotation', 'd_scaling') + (('g_rotation',) if sep else ())
                cot = {k: torch.randn(512, 4 if 'rot' in k else 3, generator=g).cuda() for k in keys}
                ok = True
                for rep in range(2):
                    got = ra.deform_network_forward(ref, x, t)
                    want = ref.reference_forward(x, t)
                    assert set(got) == set(keys) and all(rel_err(got[k], want[k]) <= 2e-5 for k in keys)
                    sh = ra.sp_net_shadow(ref)
                    assert all(a is b for a, b in zip(sh.parameters(), ref.parameters()))
                    if not sp_net_relu_masks_agree(ref, sh.runner(512), x, t):
                        ok = False
                        break
                    params = list(ref.parameters())
                    gg = torch.autograd.grad(sum((got[k] * cot[k]).sum() for k in keys), params)
                    gw = torch.autograd.grad(sum((want[k] * cot[k]).sum() for k in keys), params)
                    for (n, _), a, b in zip(ref.named_parameters(), gg, gw):
                        assert rel_err(a, b) <= 5e-5, (blender, rep, n, rel_err(a, b))
                    with torch.no_grad():   # an optimizer step in place
                        ref.linear[2].weight.add_(0.01 * torch.randn(256, 256, generator=g).cuda())
                        ref.gaussian_warp.weight.mul_(1.3)
                if ok:
                    compared += 1
                    break
        assert compared == 2, 'four draws in a row with a ReLU flip between the two fp32 evaluations'
        assert ra.calls['sp_net_fused'] >= before['sp_net_fused'] + 4
    finally:
        ra._originals.pop('sk_net', None)
        ra._originals.pop('sp_net', None)


@pytest.mark.parametrize('name', ['sk_W', 'sk_kernel', 'sp_W_LBS', 'sp_wk_LBSc_sep', 'sp_dist_LBS', 'sp_kernel_LBSc'])
def test_accelerated_calc_LBS_weight_equals_the_reference_lines(name):
    """accelerate_reference()'s ``calc_LBS_weight`` (search + weighting as one launch per direction) on a stand-in `self` with the
    inputs of the reference's own runs (sk_stage.npz): weights, indices, the side effect of sk_gs.py:771-773, and -- through a cotangent
    on the weights -- the gradients of the weighting's parameters against torch autograd of the reference's lines restated
    (ref_sequence._lbs_weights, itself replayed against the reference's recorded outputs by the tests above)"""
    import types
    L, p3d = _mods()
    from sk_gs_amd import reference_accel as ra
    z = np.load(os.path.join(GOLDEN, 'sk_stage.npz'))
    stage, K, _, _ = rs.SCENARIOS[name]
    results = []
    for fast in (True, False):
        a, out, cot, grad = rs.load_scenario(z, name, 'cuda')
        points = a['_xyz'].detach()
        bones = a['joints'] if stage == 'sk' else a['sp_points']
        feat, sfeat = (a.get('hyper_feature'), a.get('sp_hyper_feature')) if stage == 'sp' else (None, None)
        if fast:
            me = types.SimpleNamespace(num_knn=K, _sp_radius=a.get('_sp_radius'), _sp_weight=a.get('_sp_weight'), sp_W=a.get('sp_W'),
                                       sk_is_init=torch.tensor(False), sp_weights=None, sp_knn=None)
            me.kernel_radius = torch.exp(me._sp_radius) if me._sp_radius is not None else None
            me.kernel_weight = torch.sigmoid(me._sp_weight) if me._sp_weight is not None else None
            n0 = ra.calls['lbs_weight_fused']
            w, idx = ra.calc_LBS_weight(me, points, bones, feat, sfeat)
            assert ra.calls['lbs_weight_fused'] == n0 + 1
            assert torch.equal(me.sp_weights, w.detach()) and torch.equal(me.sp_knn.as_subclass(torch.Tensor), idx.as_subclass(torch.Tensor))
            assert isinstance(idx, p3d.NeighbourIndex) == p3d._TYPED_INDEX
        else:
            w, idx = rs._lbs_weights(p3d.knn_points, a, points, bones, K, feat, sfeat)
        leaves = [a[k] for k in ('_sp_radius', '_sp_weight', 'sp_W', 'hyper_feature', 'sp_hyper_feature') if k in a and a[k].requires_grad]
        names = [k for k in ('_sp_radius', '_sp_weight', 'sp_W', 'hyper_feature', 'sp_hyper_feature') if k in a and a[k].requires_grad]
        G = torch.randn(w.shape, generator=torch.Generator().manual_seed(9)).cuda()
        gr = torch.autograd.grad((w * G).sum(), leaves, allow_unused=True) if (leaves and w.requires_grad) else []
        results.append((w.detach(), idx.as_subclass(torch.Tensor), dict(zip(names, gr))))
    (w1, i1, g1), (w0, i0, g0) = results
    assert torch.equal(i1, i0) and rel_err(w1, w0) <= 2e-6
    assert torch.equal(i0.cpu(), out['_knn_i']) and rel_err(w1.cpu(), out['_knn_w']) <= 2e-6       # the reference's own run
    assert set(g1) == set(g0)
    for k in g0:
        if g0[k] is None:
            assert g1[k] is None or float(g1[k].abs().max()) == 0.0, k
        else:
            assert rel_err(g1[k], g0[k]) <= 2e-5, (k, rel_err(g1[k], g0[k]))


def test_negative_indices_wrap_and_wild_ones_address_nothing():
    """the blend / index-add launches took their int64 indices unchecked.  torch semantics now: a negative index counts from the
    end -- the same numbers as torch's own indexing, forward and backward --; an index outside [-M, M) (torch: a device assertion) reads
    and adds NOTHING instead of going through a wild LDS or global address"""
    L, p3d = _mods()
    g = torch.Generator().manual_seed(3)
    P, M, K = 5000, 20, 5
    v = torch.cat([0.4 * torch.randn(M, 3, generator=g), torch.randn(M, 4, generator=g)], -1)
    idx = torch.randint(0, M, (P, K), generator=g)
    neg = idx.clone()
    neg[::3] -= M                                                          # the same rows, addressed from the end
    pts, w0, c = torch.randn(P, 3, generator=g), torch.softmax(torch.randn(P, K, generator=g), -1), torch.randn(P, 3, generator=g)
    outs = []
    for ix in (idx, neg):
        a, w = v.cuda().requires_grad_(), w0.cuda().requires_grad_()
        d = L._se3_blend(a, ix.cuda(), pts.cuda(), w)
        (d * c.cuda()).sum().backward()
        outs.append((d.detach(), a.grad, w.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    assert rel_err(outs[0][1], outs[1][1]) <= 2e-5           # (the bones' rows are summed by LDS atomics: order-dependent bits)
    # the typed gather's backward with negative indices == torch's own index backward
    table0, cot = torch.randn(M, 4, generator=g), torch.randn(P, K, 4, generator=g).cuda()
    res = []
    for typed in (True, False):
        t = table0.cuda().requires_grad_()
        i = p3d.NeighbourIndex.wrap(neg.cuda()) if typed else neg.cuda()
        (t[i] * cot).sum().backward()
        res.append(t.grad)
    assert rel_err(res[0], res[1]) <= 2e-5
    # wild indices: skipped (rows that only hold valid ones are untouched by their neighbours' garbage)
    wild = idx.clone()
    wild[7, 2], wild[11, 0] = M + 12345, -M - 9
    a, w = v.cuda().requires_grad_(), w0.cuda().requires_grad_()
    d = L._se3_blend(a, wild.cuda(), pts.cuda(), w)
    (d * c.cuda()).sum().backward()
    keep = torch.ones(P, dtype=torch.bool)
    keep[[7, 11]] = False
    assert torch.equal(d.detach()[keep.cuda()], outs[0][0][keep.cuda()]) and bool(torch.isfinite(d).all()) and bool(torch.isfinite(a.grad).all())
    assert float(w.grad[7, 2]) == 0.0 and float(w.grad[11, 0]) == 0.0
