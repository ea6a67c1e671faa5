"""GPU tests of the superpoint stage (stage `sp`, networks/sk_gs.py:830-856): the MFMA deform network (csrc/sp_mlp.hip) against
its torch restatement (itself pinned to the reference's DeformNetwork by tests/golden/sp_deformnet.npz), the 3+8-d search and
weightings against the oracle, and the fused step against the operator path."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.nn.functional as F_

from helpers import rel_err, to_np

pytestmark = pytest.mark.gpu


def _net(seed=0, heads=0.05):
    from sk_gs_amd.superpoint import SpDeformNet
    torch.manual_seed(seed)
    net = SpDeformNet()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # outputs and gradients of a readable size (reset_parameters leaves the heads at 1e-5 / 1e-8)
        for head in (net.gaussian_warp, net.gaussian_scaling, net.gaussian_rotation):
            head.weight.normal_(0, heads, generator=g)
            head.bias.normal_(0, 0.1, generator=g)
        for layer in net.linear:
            layer.bias.normal_(0, 0.05, generator=g)
        net.timenet[0].bias.normal_(0, 0.1, generator=g)
    return net.cuda()


@pytest.mark.parametrize('M', [512, 20, 100, 1000, 1, 3])
def test_sp_net_forward_matches_the_torch_restatement(M):
    net = _net(M)
    g = torch.Generator().manual_seed(M)
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    t = torch.tensor([0.3125], device='cuda')
    run = net.runner(M)
    run.forward(x, t)
    with torch.no_grad():
        ref = net.reference_forward(x, t)
    want = torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']], 1)
    assert rel_err(run.raw, want) <= 2e-5
    u = F.normalize(ref['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda'), dim=-1)
    assert rel_err(run.bone_T, torch.cat([ref['d_xyz'], u], 1)) <= 2e-5
    assert rel_err(run.d_rot, u) <= 2e-5 and rel_err(run.d_scale, ref['d_scaling']) <= 2e-5
    # the saved activations are the restatement's hidden state
    from sk_gs_amd.superpoint import _SpNetDesc  # noqa: F401
    Mp = (M + 15) // 16 * 16
    saved = run.saved.view(torch.float32)
    Y7 = saved[Mp * 96 + 7 * Mp * 256: Mp * 96 + 8 * Mp * 256].view(Mp, 256)[:M]
    assert rel_err(Y7, ref['hidden']) <= 2e-5


@pytest.mark.parametrize('M,stage', [(512, False), (512, True), (37, True), (1000, False), (3, True)])
def test_sp_net_backward_matches_torch_autograd(M, stage):
    """every parameter gradient (8 layers, the time network, the three heads) of ONE row-block launch + ONE weight-gradient
    launch against torch autograd of the restatement; `stage`: cotangents w.r.t. (bone_T, d_rot, d_scale), the quaternion
    normalisation's backward inside the launch (sk_gs.py:847)"""
    net = _net(M + 1)
    g = torch.Generator().manual_seed(M + 7)
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    t = torch.tensor([0.7], device='cuda')
    run = net.runner(M)
    run.forward(x, t)
    ref = net.reference_forward(x, t)
    params = list(net.parameters())
    names = [n for n, _ in net.named_parameters()]
    if stage:
        g_T, g_r, g_s = (torch.randn(M, 7, generator=g).cuda(), torch.randn(M, 4, generator=g).cuda(),
                         torch.randn(M, 3, generator=g).cuda())
        u = F.normalize(ref['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda'), dim=-1)
        loss = (torch.cat([ref['d_xyz'], u], 1) * g_T).sum() + (u * g_r).sum() + (ref['d_scaling'] * g_s).sum()
    else:
        g_raw = torch.randn(M, 10, generator=g).cuda()
        loss = (torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']], 1) * g_raw).sum()
    want = torch.autograd.grad(loss, params)
    # the same gradients in fp64: a pre-activation within rounding distance of 0 takes the other side of the ReLU in ANY
    # two fp32 evaluations (torch's included) -- a parameter is held to max(5e-5, 3 x torch-fp32's own distance from fp64)
    import copy
    keep, net._runners = net._runners, {}
    net64 = copy.deepcopy(net).double()
    net._runners = keep
    ref64 = net64.reference_forward(x.double(), t.double())
    if stage:
        u64 = F.normalize(ref64['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda', dtype=torch.float64), dim=-1)
        loss64 = ((torch.cat([ref64['d_xyz'], u64], 1) * g_T.double()).sum() + (u64 * g_r.double()).sum()
                  + (ref64['d_scaling'] * g_s.double()).sum())
    else:
        loss64 = (torch.cat([ref64['d_xyz'], ref64['d_rotation'], ref64['d_scaling']], 1) * g_raw.double()).sum()
    want64 = torch.autograd.grad(loss64, list(net64.parameters()))
    for p in params:
        p.grad = torch.full_like(p, float('nan'))  # written, not accumulated
    for rep in range(2):  # (twice: the ticket of the weight-gradient launch resets itself)
        if stage:
            run.backward(g_T, g_r, g_s)
        else:
            run.backward(None, None, None, g_raw=g_raw)
    torch.cuda.synchronize()
    for n, p, w, w64 in zip(names, params, want, want64):
        assert torch.isfinite(p.grad).all(), n
        tol = max(5e-5, 3.0 * rel_err(w.double(), w64))
        assert rel_err(p.grad.double(), w64) <= tol, (n, rel_err(p.grad.double(), w64), tol)


def test_sp_net_autograd_function_and_state_dict():
    """the operator path: SpDeformNet(x, t) as an autograd node; parameters carry the reference's state_dict names"""
    net = _net(3)
    assert {'timenet.0.weight', 'timenet.2.bias', 'linear.0.weight', 'linear.5.weight', 'gaussian_warp.weight',
            'gaussian_scaling.bias', 'gaussian_rotation.weight'} <= set(net.state_dict())
    assert tuple(net.linear[5].weight.shape) == (256, 349) and tuple(net.linear[0].weight.shape) == (256, 93)
    M = 512
    x = (torch.rand(M, 3) * 2 - 1).cuda()
    t = torch.tensor([0.25], device='cuda')
    out = net(x, t)
    ref = net.reference_forward(x, t)
    gs = [torch.randn_like(out[k]) for k in ('d_xyz', 'd_rotation', 'd_scaling')]
    got = torch.autograd.grad([out[k] for k in ('d_xyz', 'd_rotation', 'd_scaling')], list(net.parameters()), gs)
    want = torch.autograd.grad([ref[k] for k in ('d_xyz', 'd_rotation', 'd_scaling')], list(net.parameters()), gs)
    for (n, _), a, b in zip(net.named_parameters(), got, want):
        assert rel_err(a, b) <= 5e-5, n
    with pytest.raises(Exception):
        net(x.cpu(), t)


def test_sp_net_called_more_than_once_per_graph():
    """two forwards (another row count, another time) before one backward -- each autograd node keeps its own activations and
    its runner; a time PER ROW (the reference's regularisers call sp_deform_net(x [M*T,3], t [M*T,1]), sk_gs.py:1383-1395) takes the
    plain-torch body, which honours it"""
    net = _net(11)
    g = torch.Generator().manual_seed(11)
    xa, xb = (torch.rand(512, 3, generator=g) * 2 - 1).cuda(), (torch.rand(96, 3, generator=g) * 2 - 1).cuda()
    ta, tb = torch.tensor([0.2], device='cuda'), torch.tensor([[0.9]], device='cuda')
    keys = ('d_xyz', 'd_rotation', 'd_scaling')
    ca = [torch.randn(512, n, generator=g).cuda() for n in (3, 4, 3)]
    cb = [torch.randn(96, n, generator=g).cuda() for n in (3, 4, 3)]
    params = list(net.parameters())

    def loss(fn):
        oa, ob = fn(xa, ta), fn(xb, tb)                       # second forward BEFORE the first backward
        oc = fn(xa, ta * 2)                                   # ... and a third one on the first runner's buffers
        return (sum((oa[k] * c).sum() for k, c in zip(keys, ca)) + sum((ob[k] * c).sum() for k, c in zip(keys, cb))
                + 0.5 * sum((oc[k] * c).sum() for k, c in zip(keys, ca)))
    got = torch.autograd.grad(loss(net), params)
    want = torch.autograd.grad(loss(net.reference_forward), params)
    for (n, _), a, b in zip(net.named_parameters(), got, want):
        assert rel_err(a, b) <= 5e-5, (n, rel_err(a, b))
    assert set(net._runners) == {512, 96}
    # one time per row
    T = 4
    x = xb[:, None, :].repeat(1, T, 1).reshape(-1, 3)
    t_rows = torch.rand(96 * T, 1, generator=g).cuda()
    out, ref = net(x, t_rows), net.reference_forward(x, t_rows)
    assert all(torch.equal(out[k], ref[k]) for k in keys)
    assert not torch.allclose(out['d_xyz'], net(x, t_rows[:1])['d_xyz'])       # the rows' own times matter


# ------------------------------------------------------------------------------------------- search + weightings
def _sp_scene(P, M, seed, F=8):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(P, 3, generator=g) * 2.6 - 1.3
    sp = pts[torch.randperm(P, generator=g)[:M]].clone() + 0.01 * torch.randn(M, 3, generator=g)
    feat = torch.full((P, F), -1e-2) + 0.03 * torch.randn(P, F, generator=g) if F else None
    sfeat = torch.full((M, F), 1e-2) + 0.03 * torch.randn(M, F, generator=g) if F else None
    radius_raw = torch.randn(M, generator=g) * 0.3 + float(np.log(0.25))
    kweight_raw = torch.randn(M, generator=g)
    return pts, sp, feat, sfeat, radius_raw, kweight_raw


def _sp_forward(P, M, K, F, pts, sp, feat, sfeat, radius_raw=None, kweight_raw=None, T=1.0, sp_W=None, order=None, hint=False,
                prev=None):
    import ctypes as C
    from sk_gs_amd import _C
    lib = _C.load_library()
    dev = 'cuda'
    idx = prev if prev is not None else torch.randint(-2 ** 40, 2 ** 40, (P, K), dtype=torch.int64, device=dev)  # (any hint is valid)
    w, d = torch.empty((P, K), device=dev), torch.empty((P, K), device=dev)
    p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    rank = None
    if order is not None and hint:
        rank = torch.empty_like(order)
        rank[order.long()] = torch.arange(M, dtype=order.dtype, device=order.device)
    _C._check(lib.skgs_sp_lbs_weights_forward(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), p(pts), p(feat), p(sp),
                                              p(sfeat), p(radius_raw), p(kweight_raw), C.c_float(T), p(sp_W), p(order), p(rank), p(idx), p(w), p(d),
                                              None, C.c_size_t(0), C.c_int32(0), C.c_int32(0), _C._stream()))
    return idx, w, d


@pytest.mark.parametrize('P,M,K,F', [(100_000, 512, 5, 8), (100_000, 512, 5, 0), (30_000, 512, 8, 8), (5_001, 100, 12, 8),
                                     (777, 7, 3, 8)])
def test_sp_search_and_weightings_match_the_oracle_at_full_size(oracle32, oracle64, P, M, K, F):
    """the 3 + 8-dimensional search over 512 superpoints at P = 100k (BASELINE config #1's Gaussians, the
    reference's num_superpoints) -- indices BIT-exact against the oracle (pytorch3d order: ascending, ties to the lower id),
    distances bit-exact, the four weightings <= 1e-6, and the backward of the distance-based ones (hyper features, superpoint
    hyper features, radii, kernel weights) <= 1e-4 of each tensor's scale against the oracle's analytic gradients."""
    import ctypes as C
    from sk_gs_amd import _C
    pts, sp, feat, sfeat, radius_raw, kweight_raw = _sp_scene(P, M, 7 * P + M, F)
    cat = lambda a, b: a if b is None else torch.cat([a, b], 1)  # noqa: E731
    want_d, want_i = oracle32.knn_bones(to_np(cat(pts, feat)), to_np(cat(sp, sfeat)), K)
    dev = 'cuda'
    d_pts, d_sp = pts.to(dev), sp.to(dev)
    d_feat, d_sfeat = (feat.to(dev), sfeat.to(dev)) if F else (None, None)
    d_rad, d_kw = radius_raw.to(dev), kweight_raw.to(dev)
    g = torch.Generator().manual_seed(P)
    G = torch.randn(P, K, generator=g)
    lib = _C.load_library()
    lib.skgs_sp_lbs_weights_workspace_bytes.restype = C.c_size_t
    ws = torch.empty((int(lib.skgs_sp_lbs_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(F))) + 16,),
                     dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    radius = np.exp(to_np(radius_raw).astype(np.float64))
    kw = 1.0 / (1.0 + np.exp(-to_np(kweight_raw).astype(np.float64)))
    for method in ('weighted_kernel', 'kernel', 'dist', 'W'):
        sp_W = torch.randn(P, M, generator=g).to(dev) if (method == 'W' and P * M <= 20_000_000) else None
        if method == 'W' and sp_W is None:
            continue
        rr = d_rad if method in ('weighted_kernel', 'kernel') else None
        kk = d_kw if method == 'weighted_kernel' else None
        T = 0.05
        idx, w, d = _sp_forward(P, M, K, F, d_pts, d_sp, d_feat, d_sfeat, rr, kk, T, sp_W)
        np.testing.assert_array_equal(to_np(idx), want_i, err_msg=method)
        np.testing.assert_array_equal(to_np(d), want_d, err_msg=method)
        # any scan order of the superpoints gives the same lists, bit for bit (a spatial one, a random one)
        from sk_gs_amd.densify import morton_order
        for order in (morton_order(d_sp).int().contiguous(), torch.randperm(M, generator=g).int().cuda()):
            idx2, w2, d2 = _sp_forward(P, M, K, F, d_pts, d_sp, d_feat, d_sfeat, rr, kk, T, sp_W, order=order)
            assert torch.equal(idx2, idx) and torch.equal(d2, d) and torch.equal(w2, w), method
            # ... and so does any starting point: a garbage hint (first call) and the previous call's own result
            idx3, w3, d3 = _sp_forward(P, M, K, F, d_pts, d_sp, d_feat, d_sfeat, rr, kk, T, sp_W, order=order, hint=True)
            assert torch.equal(idx3, idx) and torch.equal(d3, d) and torch.equal(w3, w), method
            idx4, w4, d4 = _sp_forward(P, M, K, F, d_pts, d_sp, d_feat, d_sfeat, rr, kk, T, sp_W, order=order, hint=True, prev=idx3.clone())
            assert torch.equal(idx4, idx) and torch.equal(d4, d) and torch.equal(w4, w), method
        if method == 'W':
            want_w = oracle32.lbs_weights(to_np(sp_W), want_i)
            assert np.abs(to_np(w) - want_w).max() <= 2e-6
            continue
        if method == 'dist':
            want_w, g_dist = oracle64.lbs_weights_dist(want_d, T, g_weights=to_np(G))
            g_rad = g_kw = None
        else:
            want_w, gr = oracle64.lbs_weights_kernel(want_d, want_i, radius, kw if method == 'weighted_kernel' else None,
                                                     g_weights=to_np(G))
            g_dist, g_rad, g_kw = gr['g_dist'], gr['g_radius'] * radius, gr['g_weight']
            if g_kw is not None:
                g_kw = g_kw * kw * (1 - kw)
        assert np.abs(to_np(w) - want_w).max() <= 2e-6, method
        # ---- backward
        g_feat = torch.full((P, max(F, 1)), float('nan'), device=dev)
        g_sfeat = torch.full((M, max(F, 1)), float('nan'), device=dev)
        g_r, g_k = torch.full((M,), float('nan'), device=dev), torch.full((M,), float('nan'), device=dev)
        _C._check(lib.skgs_sp_lbs_weights_backward(
            C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), p(d_feat), p(d_sfeat), p(rr), p(kk), C.c_float(T), p(w), p(idx),
            p(d), p(G.to(dev)), p(g_feat) if F else None, p(g_sfeat) if F else None, p(g_r), p(g_k), p(ws),
            C.c_size_t(ws.numel()), C.c_int32(0), _C._stream()))
        if F:
            want_gp, want_gj = oracle64.knn_dist_backward(to_np(cat(pts, feat)), to_np(cat(sp, sfeat)), want_i, g_dist)
            assert rel_err(g_feat, want_gp[:, 3:]) <= 1e-4, method
            assert rel_err(g_sfeat, want_gj[:, 3:]) <= 1e-4, method
        if g_rad is not None:
            assert rel_err(g_r, g_rad) <= 1e-4, method
        if g_kw is not None:
            assert rel_err(g_k, g_kw) <= 1e-4, method


@pytest.mark.parametrize('method', ['weighted_kernel', 'kernel', 'dist', 'W'])
def test_operator_path_at_superpoint_size_runs_the_sp_search_and_matches_the_oracle(oracle32, oracle64, method):
    """`sk_gs_amd.deform.calc_lbs_weight` -- the reference's calc_LBS_weight surface, ACTIVATED radii / kernel weights -- with a
    superpoint-sized table routes to the launches of the fused step (csrc/sp_knn.hip; 57 us where the search built for <= 60 bones
    takes 705): indices identical to the fused step's, weights <= 2e-6 of the oracle, and autograd's gradients w.r.t. the hyper
    features, the ACTIVATED radii / kernel weights and the logit table <= 1e-4 of the oracle's analytic ones."""
    from sk_gs_amd import deform
    from sk_gs_amd.deform import calc_lbs_weight
    P, M, K, F, T = 20_000, 512, 5, 8, 0.05
    pts, sp, feat, sfeat, radius_raw, kweight_raw = [t.cuda() for t in _sp_scene(P, M, 3, F)]
    g = torch.Generator().manual_seed(11)
    G = torch.randn(P, K, generator=g)
    radius, kw = torch.exp(radius_raw).requires_grad_(True), torch.sigmoid(kweight_raw).requires_grad_(True)
    feat, sfeat = feat.requires_grad_(True), sfeat.requires_grad_(True)
    sp_W = torch.randn(P, M, generator=g).cuda().requires_grad_(True) if method == 'W' else None
    calls = []
    orig = deform._SpKnnWeights.apply
    deform._SpKnnWeights.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        w, idx = calc_lbs_weight(pts, sp, K, sp_W=sp_W, kernel_radius=radius if method in ('weighted_kernel', 'kernel') else None,
                                 kernel_weight=kw if method == 'weighted_kernel' else None, temperature=T, feature=feat, sp_feature=sfeat)
    finally:
        deform._SpKnnWeights.apply = orig
    assert calls, 'the superpoint-sized route was not taken'
    (w * G.cuda()).sum().backward()
    cat = lambda a, b: torch.cat([a, b], 1)  # noqa: E731
    want_d, want_i = oracle32.knn_bones(to_np(cat(pts, feat)), to_np(cat(sp, sfeat)), K)
    np.testing.assert_array_equal(to_np(idx), want_i)
    idx_f, _, _ = _sp_forward(P, M, K, F, pts, sp, feat.detach(), sfeat.detach(), radius_raw, kweight_raw)
    assert torch.equal(idx, idx_f)  # (and the fused step's call of the same launch)
    if method == 'W':
        want_w = oracle32.lbs_weights(to_np(sp_W), want_i)
        assert np.abs(to_np(w) - want_w).max() <= 2e-6
        dot = (w.detach() * G.cuda()).sum(1, keepdim=True)
        want = torch.zeros(P, M, device='cuda').scatter_(1, idx, w.detach() * (G.cuda() - dot))
        assert rel_err(sp_W.grad, want) <= 1e-5
        assert feat.grad is None and sfeat.grad is None  # (the logits do not depend on the distances)
        return
    if method == 'dist':
        want_w, g_dist = oracle64.lbs_weights_dist(want_d, T, g_weights=to_np(G))
    else:
        want_w, gr = oracle64.lbs_weights_kernel(want_d, want_i, to_np(radius).astype(np.float64),
                                                 to_np(kw).astype(np.float64) if method == 'weighted_kernel' else None, g_weights=to_np(G))
        g_dist = gr['g_dist']
        assert rel_err(radius.grad, gr['g_radius']) <= 1e-4       # (w.r.t. the ACTIVATED radius: no d exp)
        if method == 'weighted_kernel':
            assert rel_err(kw.grad, gr['g_weight']) <= 1e-4
    assert np.abs(to_np(w) - want_w).max() <= 2e-6
    want_gp, want_gj = oracle64.knn_dist_backward(to_np(cat(pts, feat)), to_np(cat(sp, sfeat)), want_i, g_dist)
    assert rel_err(feat.grad, want_gp[:, 3:]) <= 1e-4 and rel_err(sfeat.grad, want_gj[:, 3:]) <= 1e-4


# ---------------------------------------------------------------------------------------------------- the fused step
def _sp_model(P, M, K, W, H, frames, method, seed=0, warp_method='LBS', sep_rot=False, is_blender=True, t_degree=6):
    from sk_gs_amd import scene
    from sk_gs_amd.superpoint import SuperpointGaussians
    dev = torch.device('cuda')
    model = SuperpointGaussians(P, M, K, num_frames=frames, seed=seed, scale_mult=3.0, lbs_method=method, warp_method=warp_method,
                                sep_rot=sep_rot, is_blender=is_blender, t_degree=t_degree).to(dev)
    with torch.no_grad():  # deformations large enough to matter in the image
        model.sp_deform_net.gaussian_warp.weight.mul_(20.0)
        model.sp_deform_net.gaussian_rotation.weight.mul_(20.0)
        model.sp_deform_net.gaussian_scaling.weight.mul_(100.0)
        if sep_rot:
            model.sp_deform_net.local_rotation.weight.mul_(20.0)
    cam = scene.make_camera(W, H, seed=seed)
    rs = scene.raster_settings_from_camera(cam, sh_degree=3, colmap=True, device=dev)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
    return model, rs, target


@pytest.mark.parametrize('method,warp_method,sep_rot', [('weighted_kernel', 'LBS', False), ('W', 'LBS', False), ('dist', 'LBS', False),
                                                        ('weighted_kernel', 'LBS_c', True), ('kernel', 'LBS_c', False), ('W', 'LBS', True),
                                                        ('W', 'largest', False), ('weighted_kernel', 'largest', True)])
def test_fused_superpoint_step_matches_the_operator_path(method, warp_method, sep_rot):
    """stage sp end to end: FusedSuperpointStep (straight C-ABI calls, MFMA network, one-launch 3+8-d search) against the
    autograd operator path of SuperpointGaussians.render + image_loss -- the image and EVERY parameter gradient (Gaussians,
    hyper features, superpoint tables, the network's 26 tensors; + the `local_rotation` head with sep_rot, + the superpoints' positions
    with warp_method LBS_c: the shipped SC-GS configuration is (weighted_kernel, LBS_c, sep_rot), exps/d_nerf_sc_gs.yaml:31-32)"""
    from sk_gs_amd import _C
    from sk_gs_amd.losses import image_loss
    from sk_gs_amd.superpoint import FusedSuperpointStep
    from helpers import assert_close_robust
    P, M, K, W, H, frames, tid = 6000, 512, 5, 160, 120, 3, 1
    model, rs, target = _sp_model(P, M, K, W, H, frames, method, warp_method=warp_method, sep_rot=sep_rot)
    assert ('local_rotation.weight' in dict(model.sp_deform_net.named_parameters())) == sep_rot
    _C.config.sync_num_rendered = True
    out = model.render(rs, time_id=tid)
    loss = image_loss(out['images'], target)
    loss.backward()
    ref = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
    R = out['buffer'].R
    for p in model.parameters():
        p.grad = None
    step = FusedSuperpointStep(model, W, H, capacity=int(R * 1.2) + 1024)
    # parameters this stage / weighting gives no gradient keep their zero .grad (never written): sp_points (detached
    # everywhere on this path) and, under `W`, the hyper features (the logits do not depend on the distances)
    silent = (set() if warp_method == 'LBS_c' else {'sp_points'}) | ({'hyper_feature', 'sp_hyper_feature'} if method == 'W' else set())
    for n, p in model.named_parameters():
        if n not in silent:
            p.grad.fill_(7.0)
    step.forward_backward(rs, tid, target)
    assert step.status()['overflow'] == 0
    # (the two paths bin with different tile-list layouts; 2 of 57 600 values were 1.7e-5 apart in one of the synthetic scenes)
    # (`largest`: the operator path forms the position from torch ops -- normalize, cross products -- the step from the kernel's
    # arithmetic: the means differ in the last bits and a handful of pixels by up to 5e-5)
    assert_close_robust(step.image, out['images'].detach(), 1e-4 if warp_method == 'largest' else 2e-5, 1e-4, name=f'image sp {method} {warp_method}')
    for n, p in model.named_parameters():
        if n in silent:
            assert float(p.grad.abs().max()) == 0.0 and float(ref[n].abs().max()) == 0.0, n
            continue
        assert float(ref[n].abs().max()) > 0, n
        assert_close_robust(p.grad, ref[n], 3e-4, 1e-3, name=f'{n} sp {method} {warp_method} sep_rot={sep_rot}')


@pytest.mark.parametrize('method,warp_method', [('weighted_kernel', 'LBS'), ('kernel', 'LBS'), ('dist', 'LBS'), ('W', 'LBS'), ('W', 'largest'),
                                                ('dist', 'largest')])
def test_sp_rows_pass_as_a_job_of_the_rasterizer_backward_equals_the_separate_launch(method, warp_method):
    """skgs_raster_grads.sp_skinning_job: the rows pass of the skinning + weighting backward inside the rasterizer's per-Gaussian
    backward launch (+ bones and finalize behind it) against skgs_sp_skinning_backward as a call of its own -- every gradient
    (their upstream is summed by atomics: order, not bits)"""
    from sk_gs_amd import _C
    from sk_gs_amd.superpoint import FusedSuperpointStep
    from helpers import assert_close_robust
    P, M, K, W, H, frames, tid = 6000, 512, 5, 160, 120, 3, 1
    model, rs, target = _sp_model(P, M, K, W, H, frames, method, warp_method=warp_method)
    _C.config.sync_num_rendered = True
    R = model.render(rs, time_id=tid)['buffer'].R
    got = {}
    for job in (False, True):
        for p in model.parameters():
            p.grad = None
        step = FusedSuperpointStep(model, W, H, capacity=int(R * 1.2) + 1024)
        step.deform_in_preprocess = step.deform_backward_in_preprocess = job  # (the forward's skinning likewise: deform_job)
        for p in model.parameters():
            p.grad.zero_()
        for buf in (step.means, step.scales, step.rotations, step.opacity):
            buf.fill_(-7.0)
        step.backward_raster(rs, tid, target)
        assert step._rows_backward_done == job
        step.backward_skinning(tid)
        torch.cuda.synchronize()
        got[job] = {n: p.grad.clone() for n, p in model.named_parameters()}
        got[job].update({'fwd_' + k: getattr(step, k).clone() for k in ('means', 'scales', 'rotations', 'opacity', 'radii', 'image')})
    for n, v in got[False].items():
        if n.startswith('fwd_'):  # the forward halves: the same arithmetic on the same values -- the same bits
            assert torch.equal(got[True][n], v), n
        elif float(v.abs().max()) == 0.0:
            assert float(got[True][n].abs().max()) == 0.0, n
        else:
            assert_close_robust(got[True][n], v, 2e-5, 1e-4, name=f'{n} sp {method}')


@pytest.mark.parametrize('method', ['weighted_kernel', 'W'])
def test_superpoint_train_step_graph_replay_equals_eager_steps(method):
    """FusedSuperpointTrainStep (rows' Adam on the idle CUs of the network's backward launches, closing launch for the rest; `W`:
    the dense logit table updated sparsely from the step's neighbours, no dense gradient) captured as ONE hipGraph and
    replayed = the same steps issued eagerly with a plain optimizer.step() over the dense gradients; and it trains"""
    from sk_gs_amd import _C
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.superpoint import FusedSuperpointStep, FusedSuperpointTrainStep
    from sk_gs_amd.train_step import GraphedSteps
    P, M, K, W, H, frames = 5000, 512, 5, 128, 96, 2
    runs = []
    for fused in (True, False):
        model, rs, _ = _sp_model(P, M, K, W, H, frames, method, seed=5)
        _C.config.sync_num_rendered = True
        with torch.no_grad():  # a fitting problem with a known answer: the model's own render, then perturbed colours
            first = model.render(rs, time_id=0)
            R, target = first['buffer'].R, first['images'].clamp(0, 1).contiguous()
            model._features_dc.add_(0.3 * torch.randn(model._features_dc.shape, generator=torch.Generator().manual_seed(8)).cuda())
        step = FusedSuperpointStep(model, W, H, capacity=int(R * 1.5) + 1024)
        opt = FusedAdam(model.param_groups(lr=1e-4))
        train = FusedSuperpointTrainStep(step, opt, enable=fused)
        assert train.fused == fused and step.sparse_logits == (fused and method == 'W')
        losses = []
        if fused:
            graphs = GraphedSteps(lambda _: train(rs, 0, target), collect_garbage=False)
            for i in range(6):
                graphs(0)
                losses.append(float(step.loss3[0]))
        else:
            for i in range(6):
                train(rs, 0, target)
                losses.append(float(step.loss3[0]))
        assert float(opt.step_count.item()) == 6
        runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    (la, pa), (lb, pb) = runs
    assert all(np.isfinite(la)) and la[0] > 0  # (Adam with eps 1e-15 random-walks the parameters whose gradient is noise: no monotone claim)
    for i, (a, b) in enumerate(zip(la, lb)):  # (the first step is the same arithmetic; later ones inherit the sign flips below)
        assert abs(a - b) <= (2e-5 if i == 0 else 1e-3) * max(abs(b), 1e-6), (i, a, b)
    # The loss of every step agrees to 2e-5 (above).  The parameters themselves: the blend backward sums with float atomics, a
    # gradient within rounding of zero may change sign between two runs and Adam (eps 1e-15) turns a sign into a full step --
    # tensors that start at zero (biases, _sp_weight) are ALL such elements.  Bound: no element further apart than the steps
    # taken allow (2 x steps x the largest rate), and the per-Gaussian tensors agree closely almost everywhere.
    for n in pa:
        assert float((pa[n] - pb[n]).abs().max()) <= 2 * 6 * 50 * 1e-4, n
        if pa[n].shape[0] == P and n != 'hyper_feature':
            scale = float(pb[n].abs().max().clamp_min(1e-30))
            off = ((pa[n] - pb[n]).abs() > 2e-5 * scale).float().mean()
            assert float(off) <= 3e-2, (n, float(off))  # (run to run 0.1 % .. 1 % of the elements: the atomics' order decides)


@pytest.mark.parametrize('P,M,K,F,method', [(100_000, 512, 5, 8, 'weighted_kernel'), (20_000, 512, 5, 8, 'dist'),
                                            (20_000, 512, 5, 8, 'W'), (3_000, 100, 7, 0, 'kernel'), (257, 9, 3, 8, 'weighted_kernel')])
def test_sp_skinning_backward_by_inverse_lists_matches_the_two_call_sequence_and_the_oracle(oracle32, P, M, K, F, method):
    """skgs_sp_skinning_backward (rows | bones over the inverse neighbour lists | finalize: no atomics) against the sequence it
    replaces (skgs_lbs_deform_backward + skgs_sp_lbs_weights_backward, LDS atomics) on every output, and its skinning part
    against the oracle (lbs_deform_backward) at P = 100k, M = 512"""
    import ctypes as C
    from sk_gs_amd import _C, scene
    lib = _C.load_library()
    dev = 'cuda'
    pts, sp, feat, sfeat, radius_raw, kweight_raw = _sp_scene(P, M, P + 3 * M, F)
    gs = scene.make_gaussians(P, seed=2)
    gen = torch.Generator().manual_seed(P + 1)
    spT = torch.cat([0.05 * torch.randn(M, 3, generator=gen),
                     F_.normalize(0.2 * torch.randn(M, 4, generator=gen) + torch.tensor([0, 0, 0, 1.]), dim=-1)], 1)
    d_rot, d_scale = spT[:, 3:].clone(), 0.01 * torch.randn(M, 3, generator=gen)
    cu = lambda t: None if t is None else t.to(dev).contiguous()  # noqa: E731
    d_pts, d_sp, d_feat, d_sfeat = cu(pts), cu(sp), cu(feat), cu(sfeat)
    rr = cu(radius_raw) if method in ('weighted_kernel', 'kernel') else None
    kk = cu(kweight_raw) if method == 'weighted_kernel' else None
    sp_W = torch.randn(P, M, generator=gen).to(dev) if method == 'W' else None
    T = 0.07
    p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    for fn in (lib.skgs_sp_pairs_bytes, lib.skgs_sp_skinning_backward_workspace_bytes, lib.skgs_sp_lbs_weights_workspace_bytes,
               lib.skgs_lbs_deform_backward_workspace_bytes):
        fn.restype = C.c_size_t
    pairs = torch.zeros((int(lib.skgs_sp_pairs_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(K))),), dtype=torch.uint8, device=dev)
    idx = torch.empty((P, K), dtype=torch.int64, device=dev)
    w, dist = torch.empty((P, K), device=dev), torch.empty((P, K), device=dev)
    for rep in range(2):  # (twice: the forward clears the lists it filed before)
        _C._check(lib.skgs_sp_lbs_weights_forward(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), p(d_pts), p(d_feat), p(d_sp),
                                                  p(d_sfeat), p(rr), p(kk), C.c_float(T), p(sp_W), None, None, p(idx), p(w), p(dist), p(pairs),
                                                  C.c_size_t(pairs.numel()), C.c_int32(0), C.c_int32(0), _C._stream()))
    torch.cuda.synchronize()
    hdr = pairs[:8].view(torch.int32)
    counts = pairs[256:256 + 4 * M].view(torch.int32)
    assert int(hdr[1]) == 0 and int(counts.sum()) == P * K
    assert torch.equal(counts.long(), torch.bincount(idx.view(-1), minlength=M))
    xyz, ls, rot, op = cu(gs['xyz']), cu(gs['log_scale']), cu(gs['rot']), cu(gs['opacity_logit'])
    d = _C._DeformInputs()
    d.P, d.K, d.M = P, K, M
    d.points = d.xyz = d_pts.data_ptr()
    c_T, c_rot, c_scale = cu(spT), cu(d_rot), cu(d_scale)
    d.weights, d.indices, d.bone_T, d.bone_drot, d.bone_dscale = w.data_ptr(), idx.data_ptr(), c_T.data_ptr(), c_rot.data_ptr(), c_scale.data_ptr()
    d.log_scale, d.rot, d.opacity_logit = ls.data_ptr(), rot.data_ptr(), op.data_ptr()
    g_means, g_scales = torch.randn(P, 3, generator=gen).to(dev), torch.randn(P, 3, generator=gen).to(dev)
    g_rots, g_op = torch.randn(P, 4, generator=gen).to(dev), torch.randn(P, 1, generator=gen).to(dev)
    nan = lambda *s_: torch.full(s_, float('nan'), device=dev)  # noqa: E731

    def outs():
        return dict(g_w=nan(P, K), g_xyz=nan(P, 3), g_ls=nan(P, 3), g_rot=nan(P, 4), g_op=nan(P, 1), g_feat=nan(P, max(F, 1)),
                    g_T=nan(M, 7), g_dr=nan(M, 4), g_ds=nan(M, 3), g_sf=nan(M, max(F, 1)), g_r=nan(M), g_k=nan(M))
    a, b = outs(), outs()
    ws = torch.empty((int(lib.skgs_sp_skinning_backward_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(K))),), dtype=torch.uint8, device=dev)
    logits = method == 'W'
    _C._check(lib.skgs_sp_skinning_backward(
        C.byref(d), C.c_int32(F), p(d_feat), p(d_sfeat), p(rr), p(kk), C.c_float(T), C.c_int32(int(logits)), p(dist), p(g_means),
        p(g_scales), p(g_rots), p(g_op), p(a['g_w']), p(a['g_xyz']), p(a['g_ls']), p(a['g_rot']), p(a['g_op']),
        p(a['g_feat']) if F and not logits else None, p(a['g_T']), p(a['g_dr']), p(a['g_ds']), p(a['g_sf']) if F and not logits else None,
        p(a['g_r']) if rr is not None else None, p(a['g_k']) if kk is not None else None, p(pairs), C.c_size_t(pairs.numel()), p(ws),
        C.c_size_t(ws.numel()), _C._stream()))
    # ---- the two-call sequence it replaces
    dws = torch.empty((int(lib.skgs_lbs_deform_backward_workspace_bytes(C.c_int32(P), C.c_int32(M))),), dtype=torch.uint8, device=dev)
    _C._check(lib.skgs_lbs_deform_backward(C.byref(d), p(g_means), p(g_scales), p(g_rots), p(g_op), p(b['g_w']), p(b['g_T']), p(b['g_dr']),
                                           p(b['g_ds']), p(b['g_xyz']), p(b['g_ls']), p(b['g_rot']), p(b['g_op']), p(dws),
                                           C.c_size_t(dws.numel()), _C._stream()))
    names = ['g_w', 'g_xyz', 'g_ls', 'g_rot', 'g_op', 'g_T', 'g_dr', 'g_ds']
    if not logits:
        sws = torch.empty((int(lib.skgs_sp_lbs_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(F))) + 16,), dtype=torch.uint8, device=dev)
        _C._check(lib.skgs_sp_lbs_weights_backward(
            C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), p(d_feat), p(d_sfeat), p(rr), p(kk), C.c_float(T), p(w), p(idx),
            p(dist), p(b['g_w']), p(b['g_feat']) if F else None, p(b['g_sf']) if F else None, p(b['g_r']), p(b['g_k']), p(sws),
            C.c_size_t(sws.numel()), C.c_int32(0), _C._stream()))
        names += (['g_feat', 'g_sf'] if F else []) + (['g_r'] if rr is not None else []) + (['g_k'] if kk is not None else [])
    torch.cuda.synchronize()
    for nme in names:
        assert torch.isfinite(a[nme]).all(), nme
        assert rel_err(a[nme], b[nme]) <= 2e-5, (nme, rel_err(a[nme], b[nme]))
    # the skinning part against the oracle
    want = oracle32.lbs_deform_backward(to_np(pts), to_np(w), to_np(idx), to_np(spT), to_np(d_rot), to_np(d_scale), to_np(gs['log_scale']),
                                        to_np(gs['rot']), to_np(gs['opacity_logit']), to_np(g_means), to_np(g_scales), to_np(g_rots),
                                        to_np(g_op))
    for nme, key in (('g_w', 'g_weights'), ('g_T', 'g_bone_T'), ('g_dr', 'g_bone_drot'), ('g_ds', 'g_bone_dscale'), ('g_xyz', 'g_xyz'),
                     ('g_rot', 'g_rot')):
        assert rel_err(a[nme], want[key]) <= 1e-4, (nme, rel_err(a[nme], want[key]))
    assert int(counts.sum()) == 0  # consumed: the finalize launch cleared the lists


@pytest.mark.parametrize('method', ['weighted_kernel', 'W'])
def test_superpoint_training_example_runs_and_fits(method):
    """examples/train_superpoints.py: the stage-sp loop a user would write (one captured graph for all views, fused optimizer,
    `W`: the sparse logit-table update) runs, stays finite, overflows nothing and lowers the loss"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_superpoints.py'), '--gaussians', '6000', '--size', '128',
                        '--iters', '120', '--views', '4', '--lbs-method', method], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    last = p.stdout.strip().splitlines()[-1]
    assert f'LBS_method {method}' in last and f'sparse logit update: {method == "W"}' in last, last
    first, final = [float(x) for x in last.split('loss ')[1].split(',')[0].split(' -> ')]
    assert final < 0.8 * first, last


def test_inverse_list_overflow_is_reported_and_sticky():
    """a superpoint whose inverse list outgrows its capacity (16 x the mean list) loses pairs in the backward's sums; the
    per-step flag is cleared by the next forward's preparation, the EVENT counter is not -- a loop that asks every N steps sees it"""
    from sk_gs_amd import _C
    from sk_gs_amd.superpoint import FusedSuperpointStep
    P, M, K, W, H = 40_000, 64, 2, 96, 64        # mean list 1250 -> capacity 20 000 < P
    model, rs, target = _sp_model(P, M, K, W, H, 2, 'dist')
    _C.config.sync_num_rendered = True
    R = model.render(rs, time_id=0)['buffer'].R
    for p in model.parameters():
        p.grad = None
    step = FusedSuperpointStep(model, W, H, capacity=int(R * 3) + 4096)
    step.forward_backward(rs, 0, target)
    st = step.status()
    assert st['pairs_overflow'] == 0 and st['pairs_overflow_events'] == 0
    keep = model._xyz.detach().clone()
    with torch.no_grad():  # every Gaussian next to superpoint 0: its list would hold all P of them
        model._xyz.copy_(model.sp_points[0] + 1e-3 * torch.randn(P, 3, device='cuda'))
        if model.hyper_feature is not None:
            model.hyper_feature.copy_(model.sp_hyper_feature[0].expand(P, -1))
    step.forward_backward(rs, 0, target)
    st = step.status()
    assert st['pairs_overflow'] == 1 and st['pairs_overflow_events'] == 1
    step.forward_backward(rs, 1, target)
    assert step.status()['pairs_overflow_events'] == 2
    with torch.no_grad():
        model._xyz.copy_(keep)
    step.forward_backward(rs, 0, target)          # a clean step: the flag of the last step is clear, the events remain
    st = step.status()
    assert st['pairs_overflow'] == 0 and st['pairs_overflow_events'] == 2


def test_restored_checkpoint_keeps_the_sparse_logit_update_equal_to_the_dense_one():
    """the sparse `W` logit update visits only tiles whose moments are live according to its mask; a checkpoint restored into an
    existing train step (FusedAdam.load_state_dict) brings live moments in tiles the CURRENT neighbours do not touch -- the optimizer now
    tells the step (add_state_listener) and the mask is rebuilt, so those rows keep decaying exactly as under dense Adam"""
    from sk_gs_amd import _C
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.superpoint import FusedSuperpointStep, FusedSuperpointTrainStep
    P, M, K, W, H, frames = 4000, 512, 5, 96, 64, 2

    def build(sparse):
        model, rs, target = _sp_model(P, M, K, W, H, frames, 'W', seed=9)
        _C.config.sync_num_rendered = True
        with torch.no_grad():
            R = model.render(rs, time_id=0)['buffer'].R
        step = FusedSuperpointStep(model, W, H, capacity=int(R * 3) + 4096)
        opt = FusedAdam(model.param_groups(lr=1e-3))
        train = FusedSuperpointTrainStep(step, opt, sparse_logits=sparse)
        assert step.sparse_logits == sparse
        return model, rs, target, step, opt, train

    model, rs, target, step, opt, train = build(False)          # the donor: three dense steps
    for _ in range(3):
        train(rs, 0, target)
    sd = opt.state_dict()
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    got = {}
    for sparse in (True, False):
        model, rs, target, step, opt, train = build(sparse)
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(params[n])
            model._xyz.add_(0.6)                                 # other neighbours than the donor's: its live tiles are not visited now
        opt.load_state_dict(sd)
        if sparse:
            st = opt.state[model.sp_W]
            live = (st['exp_avg'].view(P, M // 32, 32) != 0).any(-1)
            mask_bits = torch.stack([(step.logit_mask >> t) & 1 for t in range(M // 32)], 1).bool()
            assert bool((mask_bits | ~live).all()), 'the mask must cover every tile with a live moment after the restore'
        for _ in range(2):
            train(rs, 1, target)
        torch.cuda.synchronize()
        got[sparse] = (model.sp_W.detach().clone(), opt.state[model.sp_W]['exp_avg'].clone(), opt.state[model.sp_W]['exp_avg_sq'].clone())
    for a, b, name in zip(got[True], got[False], ('sp_W', 'exp_avg', 'exp_avg_sq')):
        # (two runs differ where the blend backward's atomics decide a rounding; a tile that stopped being updated differs by whole steps)
        far = ((a - b).abs() > 1e-5 * float(b.abs().max())).float().mean()
        assert float(far) <= 1e-3, (name, float(far))


@pytest.mark.parametrize('M,sep_rot,lbs_c', [(512, True, False), (512, False, True), (100, True, True), (37, True, True)])
def test_sp_net_local_rotation_head_and_recentring(M, sep_rot, lbs_c):
    """the network kernels with the `local_rotation` head (sep_rot, sk_gs.py:275-282,315,848: a raw row of 14) and with the
    re-centring of warp_method LBS_c in the epilogue (bone_T.t = d_xyz + x + R(u)(-x), sk_gs.py:803-804) against torch autograd of the
    restatement: raw outputs, bone_T / d_rot / d_scale, every parameter gradient and d loss / d sp_points"""
    import copy
    from sk_gs_amd.skeleton import quat_act
    from sk_gs_amd.superpoint import SpDeformNet
    torch.manual_seed(M)
    net = SpDeformNet(sep_rot=sep_rot)
    g = torch.Generator().manual_seed(M + 1)
    with torch.no_grad():
        heads = [net.gaussian_warp, net.gaussian_scaling, net.gaussian_rotation] + ([net.local_rotation] if sep_rot else [])
        for head in heads:
            head.weight.normal_(0, 0.05, generator=g)
            head.bias.normal_(0, 0.1, generator=g)
    net = net.cuda()
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda().requires_grad_()
    t = torch.tensor([0.4], device='cuda')
    bias = torch.tensor([0, 0, 0, 1.], device='cuda')
    net64 = copy.deepcopy(net).double()          # (before the first runner exists: its ctypes tables do not copy)
    run = net.runner(M, lbs_c=lbs_c)
    run.forward(x.detach(), t)
    x64 = x.detach().double().requires_grad_()

    def stage(n_, xx):
        ref = n_.reference_forward(xx.detach(), t.to(xx.dtype))
        u = F.normalize(ref['d_rotation'] + bias.to(xx.dtype), dim=-1)
        blend = F.normalize(ref['g_rotation'] + bias.to(xx.dtype), dim=-1) if sep_rot else u
        tt = ref['d_xyz'] + xx + quat_act(u, -xx) if lbs_c else ref['d_xyz']
        return ref, torch.cat([tt, u], 1), blend, ref['d_scaling']
    ref, bone_T, blend, d_scale = stage(net, x)
    raw = torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']] + ([ref['g_rotation']] if sep_rot else []), 1)
    assert run.raw.shape == (M, 14 if sep_rot else 10) and rel_err(run.raw, raw) <= 2e-5
    assert rel_err(run.bone_T, bone_T) <= 2e-5 and rel_err(run.d_rot, blend) <= 2e-5 and rel_err(run.d_scale, d_scale) <= 2e-5
    g_T, g_r, g_s = torch.randn(M, 7, generator=g).cuda(), torch.randn(M, 4, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    _, bT64, bl64, ds64 = stage(net64, x64)
    loss64 = (bT64 * g_T.double()).sum() + (bl64 * g_r.double()).sum() + (ds64 * g_s.double()).sum()
    params64 = list(net64.parameters())
    want64 = torch.autograd.grad(loss64, params64 + ([x64] if lbs_c else []))
    loss32 = (bone_T * g_T).sum() + (blend * g_r).sum() + (d_scale * g_s).sum()
    params = list(net.parameters())
    want32 = torch.autograd.grad(loss32, params + ([x] if lbs_c else []), retain_graph=True)
    for p in params:
        p.grad = torch.full_like(p, float('nan'))
    g_x = torch.full((M, 3), float('nan'), device='cuda')
    run.backward(g_T, g_r, g_s, g_points=g_x if lbs_c else None)
    torch.cuda.synchronize()
    names = [n for n, _ in net.named_parameters()]
    for n, p, w, w64 in zip(names, params, want32, want64):
        assert torch.isfinite(p.grad).all(), n
        tol = max(5e-5, 3.0 * rel_err(w.double(), w64))     # (ReLU pre-activations within rounding of 0: see the test above)
        assert rel_err(p.grad.double(), w64) <= tol, (n, rel_err(p.grad.double(), w64), tol)
    if lbs_c:
        assert rel_err(g_x.double(), want64[-1]) <= 5e-5, rel_err(g_x.double(), want64[-1])
    # the autograd operator returns the fourth output and routes its cotangent
    if sep_rot:
        out = net(x.detach(), t)
        assert set(out) == {'d_xyz', 'd_rotation', 'd_scaling', 'g_rotation'} and rel_err(out['g_rotation'], ref['g_rotation']) <= 2e-5
        c = torch.randn(M, 4, generator=g).cuda()
        got = torch.autograd.grad((out['g_rotation'] * c).sum(), params, allow_unused=True)
        want = torch.autograd.grad((ref['g_rotation'] * c).sum(), params, allow_unused=True, retain_graph=True)
        for n, a, b in zip(names, got, want):
            if b is not None and float(b.abs().max()) > 0:
                assert rel_err(a, b) <= 5e-5, n


@pytest.mark.parametrize('method,warp_method,sep_rot,K', [('weighted_kernel', 'LBS_c', True, 3), ('W', 'largest', False, 3), ('dist', 'LBS', True, 5),
                                                          ('kernel', 'LBS_c', False, 4)])
def test_operator_path_variants_equal_the_reference_sequence(method, warp_method, sep_rot, K):
    """SuperpointGaussians.forward with the warp / network variants of the shipped configs (exps/d_nerf_sc_gs.yaml, d_nerf_sp_gs.yaml)
    against the reference's call sequence on the stand-ins (benchlib/ref_sequence.py, itself pinned by the reference's own run in
    tests/golden/sk_stage.npz): the deformed Gaussians and every gradient that reaches a parameter of the stage"""
    from benchlib import ref_sequence as rs
    from sk_gs_amd import lietorch as L, pytorch3d_ops as p3d
    P, M = 20_000, 512
    model, _, _ = _sp_model(P, M, K, 64, 64, 2, method, seed=3, warp_method=warp_method, sep_rot=sep_rot)
    tid = 1
    g = torch.Generator().manual_seed(5)
    cot = {k: torch.randn(P, n, generator=g).cuda() for k, n in (('points', 3), ('scales', 3), ('rotations', 4), ('opacity', 1))}
    out = model(tid)
    sum((out[k] * cot[k]).sum() for k in cot).backward()
    got = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    net = model.sp_deform_net
    ref = net.reference_forward(model.sp_points.detach(), model.frame_times[tid])
    a = {'_xyz': model._xyz, '_scaling': model._scaling, '_rotation': model._rotation, '_opacity': model._opacity, 'sp_points': model.sp_points,
         'hyper_feature': model.hyper_feature, 'sp_hyper_feature': model.sp_hyper_feature, 'net_d_xyz': ref['d_xyz'],
         'net_d_rotation': ref['d_rotation'], 'net_d_scaling': ref['d_scaling']}
    if sep_rot:
        a['net_g_rotation'] = ref['g_rotation']
    for k in ('_sp_radius', '_sp_weight', 'sp_W'):
        if getattr(model, k) is not None:
            a[k] = getattr(model, k)
    res = rs.sp_stage(L, p3d.knn_points, a, K, warp_method, sep_rot)
    sum((res[k] * cot[k]).sum() for k in cot).backward()
    for k in cot:
        assert rel_err(out[k], res[k]) <= 5e-5, k     # (the network's kernels against its torch body: 2e-5 on the raw outputs)
    if warp_method == 'largest':
        assert torch.equal(model.p2sp, res['p2sp'])
    checked = 0
    for n, p in model.named_parameters():
        if p.grad is None:
            assert got[n] is None or float(got[n].abs().max()) == 0.0, n
            continue
        assert got[n] is not None, n
        tol = 2e-4 if n.startswith('sp_deform_net') else 3e-5      # (the network's kernels against its torch body: ReLU flips)
        assert rel_err(got[n], p.grad) <= tol, (n, rel_err(got[n], p.grad))
        checked += 1
    assert checked >= 8 + (1 if warp_method == 'LBS_c' else 0)


from helpers import sp_net_relu_masks_agree as _relu_masks_agree  # noqa: E402


@pytest.mark.parametrize('M,t_degree,sep_rot,lbs_c', [(512, 10, False, False), (512, 6, True, True), (37, 10, True, False), (100, 15, False, True),
                                                      (3, 0, False, False)])
def test_sp_net_without_the_time_branch(M, t_degree, sep_rot, lbs_c):
    """VERDICT r4 #3, last variant: ``DeformNetwork(is_blender=False)`` (sk_gs.py:220,255-261: no time network, ``t_emb = freq(t)``,
    84 input columns at the degree 10 its comment names) through the SAME kernels (``SKGS_SP_NET_RAW_TIME_DEGREE``): raw outputs, the
    stage's epilogue and every parameter gradient against torch autograd of the restatement (pinned to the reference's class by
    tests/golden/sp_deformnet.npz ``raw32``).  Gradients are compared on a draw in which both evaluations took the same side of every
    ReLU (``_relu_masks_agree``; at most a few seeds are needed)"""
    from sk_gs_amd.skeleton import quat_act
    from sk_gs_amd.superpoint import SpDeformNet
    bias = torch.tensor([0, 0, 0, 1.], device='cuda')
    compared = False
    for seed in range(4):
        torch.manual_seed(1000 * seed + M + t_degree)
        net = SpDeformNet(sep_rot=sep_rot, is_blender=False, t_degree=t_degree)
        assert net.in_dim == 64 + 2 * t_degree and tuple(net.linear[5].weight.shape) == (256, 256 + net.in_dim) and not hasattr(net, 'timenet')
        assert net.kernel_supported()
        g = torch.Generator().manual_seed(1000 * seed + M + 1)
        with torch.no_grad():
            heads = [net.gaussian_warp, net.gaussian_scaling, net.gaussian_rotation] + ([net.local_rotation] if sep_rot else [])
            for head in heads:
                head.weight.normal_(0, 0.05, generator=g)
                head.bias.normal_(0, 0.1, generator=g)
            for layer in net.linear:
                layer.bias.normal_(0, 0.05, generator=g)
        net = net.cuda()
        x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda().requires_grad_()
        t = torch.tensor([0.4375], device='cuda')
        run = net.runner(M, lbs_c=lbs_c)
        run.forward(x.detach(), t)

        ref = net.reference_forward(x.detach(), t)
        u = F.normalize(ref['d_rotation'] + bias, dim=-1)
        blend = F.normalize(ref['g_rotation'] + bias, dim=-1) if sep_rot else u
        bone_T = torch.cat([ref['d_xyz'] + x + quat_act(u, -x) if lbs_c else ref['d_xyz'], u], 1)
        d_scale = ref['d_scaling']
        raw = torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']] + ([ref['g_rotation']] if sep_rot else []), 1)
        assert rel_err(run.raw, raw) <= 2e-5, rel_err(run.raw, raw)
        assert rel_err(run.bone_T, bone_T) <= 2e-5 and rel_err(run.d_rot, blend) <= 2e-5 and rel_err(run.d_scale, d_scale) <= 2e-5
        if not _relu_masks_agree(net, run, x.detach(), t):
            continue
        g_T, g_r, g_s = torch.randn(M, 7, generator=g).cuda(), torch.randn(M, 4, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
        loss = (bone_T * g_T).sum() + (blend * g_r).sum() + (d_scale * g_s).sum()
        params = list(net.parameters())
        want = torch.autograd.grad(loss, params + ([x] if lbs_c else []), retain_graph=True)
        for p in params:
            p.grad = torch.full_like(p, float('nan'))
        g_x = torch.full((M, 3), float('nan'), device='cuda')
        for _ in range(2):
            run.backward(g_T, g_r, g_s, g_points=g_x if lbs_c else None)
        torch.cuda.synchronize()
        names = [n for n, _ in net.named_parameters()]
        for n, p, w in zip(names, params, want):
            assert torch.isfinite(p.grad).all(), n
            assert rel_err(p.grad, w) <= 5e-5, (n, rel_err(p.grad, w))
        if lbs_c:
            assert rel_err(g_x, want[-1]) <= 5e-5
        # the autograd operator
        out = net(x.detach(), t)
        keys = ('d_xyz', 'd_rotation', 'd_scaling') + (('g_rotation',) if sep_rot else ())
        gs = [torch.randn(M, out[k].shape[1], generator=g).cuda() for k in keys]
        got = torch.autograd.grad([out[k] for k in keys], params, gs)
        want = torch.autograd.grad([ref[k] for k in keys], params, gs)
        for n, a, b in zip(names, got, want):
            assert rel_err(a, b) <= 5e-5, (n, rel_err(a, b))
        compared = True
        break
    assert compared, 'four draws in a row with a ReLU flip between the two fp32 evaluations'


def test_fused_superpoint_step_without_the_time_branch_and_its_time_noise():
    """is_blender=False end to end: with the noise off the fused step equals the operator path (image, every gradient); with it on
    (sk_gs.py:837-839: t + randn * time_interval * get_smooth_scale()) a captured step draws a FRESH time per replay and the spread of
    the network's outputs follows the scale; ``smooth_scale`` restates sk_gs.py:723-740"""
    from sk_gs_amd import _C
    from sk_gs_amd.losses import image_loss
    from sk_gs_amd.superpoint import FusedSuperpointStep, SuperpointGaussians
    from helpers import assert_close_robust
    P, M, K, W, H, frames, tid = 6000, 512, 5, 160, 120, 3, 1
    model, rs, target = _sp_model(P, M, K, W, H, frames, 'weighted_kernel', warp_method='LBS_c', sep_rot=True, is_blender=False, t_degree=10)
    _C.config.sync_num_rendered = True
    out = model.render(rs, time_id=tid)
    image_loss(out['images'], target).backward()
    ref = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
    R = out['buffer'].R
    for p in model.parameters():
        p.grad = None
    step = FusedSuperpointStep(model, W, H, capacity=int(R * 1.2) + 1024)
    step.forward_backward(rs, tid, target)
    assert step.status()['overflow'] == 0
    # (the network's kernels against its torch body differ by ~1e-6 in the superpoint transforms; the high time frequencies make that
    # visible in a couple of pixels: 2 of 57 600 at 7.5e-5 -- the north-star tolerance applies)
    assert_close_robust(step.image, out['images'].detach(), 1e-4, 1e-4, name='image sp raw time')
    for n, p in model.named_parameters():
        if n in ('hyper_feature', 'sp_hyper_feature') and float(ref[n].abs().max()) == 0:
            continue
        assert_close_robust(p.grad, ref[n], 3e-4, 1e-3, name=f'{n} sp raw time')
    # ---- the noise: a captured forward draws a new time per replay
    quiet = step.net.bone_T.clone()
    step.set_time_noise(0.05)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step.forward(rs, tid)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            step.forward(rs, tid)
    draws = []
    for _ in range(6):
        graph.replay()
        torch.cuda.synchronize()
        draws.append(step.net.bone_T.clone())
    d = torch.stack(draws)
    assert float((d[1:] - d[:-1]).abs().amax(dim=(1, 2)).min()) > 0, 'every replay must see another time'
    assert bool(torch.isfinite(d).all())
    step.set_time_noise(0.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(step.net.bone_T, quiet)
    # ---- the annealing (sk_gs.py:723-740)
    f = SuperpointGaussians.smooth_scale
    assert f(-1, 0.1, 1000) == 0.0 and f(5, 0.0, 1000, lr_final=0.0) == 0.0
    assert abs(f(0, 0.1, 1000) - 0.1) < 1e-12 and abs(f(500, 0.1, 1000) - (0.05 + 0.5e-15)) < 1e-12 and abs(f(2000, 0.1, 1000) - 1e-15) < 1e-20
    # the operator path draws its noise with torch.randn_like
    model.time_noise = 0.05
    a, b = model.superpoint_transforms(tid)[0], model.superpoint_transforms(tid)[0]
    assert float((a - b).detach().abs().max()) > 0
