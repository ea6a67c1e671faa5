"""CPU tests: every backward of the oracle is checked against central finite differences of its own forward in the
fp64 twin (oracle/skgs_oracle.c compiled with -DSKGS_F64).  This pins the backward restatements independently of
any GPU: the reference's backward is the analytic derivative of its forward (except where noted)."""
import numpy as np
import pytest
import torch

from helpers import scene_inputs, to_np


def _loss_fwd(o, act, rs, gcol, gop):
    n = to_np
    fwd = o.rasterize_forward(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.scale_modifier,
                              rs.colmap, n(rs.viewmatrix), n(rs.projmatrix), n(rs.campos), act['means3D'], act['opacity'],
                              act['sh'], act['scales'], act['rotations'])
    return float((fwd['color'] * gcol).sum() + (fwd['opacity'] * gop).sum()), fwd


def test_rasterizer_backward_matches_finite_differences_colmap(oracle64):
    P, W, H = 40, 48, 48
    act_t, rs, cam = scene_inputs(P, W, H, seed=5, colmap=True, scale_mult=12.0)
    act = {k: to_np(v).astype(np.float64) for k, v in act_t.items()}
    rng = np.random.RandomState(0)
    gcol, gop = rng.randn(3, H, W), rng.randn(H, W)
    L0, fwd = _loss_fwd(oracle64, act, rs, gcol, gop)
    n = to_np
    g = oracle64.rasterize_backward(fwd, H, W, rs.tanfovx, rs.tanfovy, rs.sh_degree, 1.0, True, n(rs.viewmatrix),
                                    n(rs.projmatrix), n(rs.campos), act['means3D'], act['sh'], act['scales'],
                                    act['rotations'], gcol, gop)
    pairs = [('means3D', 'dL_dmeans3D'), ('scales', 'dL_dscales'), ('rotations', 'dL_drotations'),
             ('opacity', 'dL_dopacity'), ('sh', 'dL_dsh')]
    vis = np.nonzero(fwd['radii'] > 0)[0]
    assert len(vis) > 10
    ok = tot = 0
    for name, gname in pairs:
        for _ in range(12):
            i = int(rng.choice(vis))
            idx = (i,) + tuple(rng.randint(0, s) for s in act[name].shape[1:])
            eps = 1e-6 * max(1.0, abs(act[name][idx]))
            a_p = {k: v.copy() for k, v in act.items()}
            a_m = {k: v.copy() for k, v in act.items()}
            a_p[name][idx] += eps
            a_m[name][idx] -= eps
            fd = (_loss_fwd(oracle64, a_p, rs, gcol, gop)[0] - _loss_fwd(oracle64, a_m, rs, gcol, gop)[0]) / (2 * eps)
            an = g[gname].reshape(act[name].shape)[idx]
            tot += 1
            # thresholds (alpha < 1/255, T < 1e-4, radius/tile changes) make the loss piecewise smooth: allow a few misses
            if abs(fd - an) <= 1e-4 * max(1.0, abs(fd), abs(an)):
                ok += 1
    assert ok >= 0.9 * tot, (ok, tot)


def test_deform_backward_matches_finite_differences(oracle64):
    from sk_gs_amd import scene
    P, M, K = 30, 6, 3
    g0 = scene.make_gaussians(P, seed=1)
    b = scene.make_bones(M, seed=1)
    gen = torch.Generator().manual_seed(4)
    ins = dict(
        weights=torch.softmax(torch.randn(P, K, generator=gen), -1), bone_T=torch.cat(
            [0.3 * torch.randn(M, 3, generator=gen), torch.randn(M, 4, generator=gen)], -1),
        bone_drot=0.1 * torch.randn(M, 4, generator=gen), bone_dscale=0.05 * torch.randn(M, 3, generator=gen),
        xyz=g0['xyz'], log_scale=g0['log_scale'], rot=g0['rot'], opacity_logit=g0['opacity_logit'])
    ins = {k: to_np(v).astype(np.float64) for k, v in ins.items()}
    idx = np.stack([np.random.RandomState(i).permutation(M)[:K] for i in range(P)]).astype(np.int64)
    rng = np.random.RandomState(3)
    cot = dict(means=rng.randn(P, 3), scales=rng.randn(P, 3), rotations=rng.randn(P, 4), opacity=rng.randn(P, 1))
    points = ins['xyz'].copy()  # the detached copy: held fixed while xyz is perturbed

    def loss(d):
        o = oracle64.lbs_deform_forward(points, d['weights'], idx, d['bone_T'], d['bone_drot'], d['bone_dscale'],
                                        d['xyz'], d['log_scale'], d['rot'], d['opacity_logit'])
        return sum(float((o[k] * cot[k]).sum()) for k in cot)

    g = oracle64.lbs_deform_backward(points, ins['weights'], idx, ins['bone_T'], ins['bone_drot'], ins['bone_dscale'],
                                     ins['log_scale'], ins['rot'], ins['opacity_logit'], cot['means'], cot['scales'],
                                     cot['rotations'], cot['opacity'])
    names = dict(weights='g_weights', bone_T='g_bone_T', bone_drot='g_bone_drot', bone_dscale='g_bone_dscale',
                 xyz='g_xyz', log_scale='g_log_scale', rot='g_rot', opacity_logit='g_opacity_logit')
    for name, gname in names.items():
        for _ in range(10):
            ix = tuple(rng.randint(0, s) for s in ins[name].shape)
            eps = 1e-6
            p, m = {k: v.copy() for k, v in ins.items()}, {k: v.copy() for k, v in ins.items()}
            p[name][ix] += eps
            m[name][ix] -= eps
            fd = (loss(p) - loss(m)) / (2 * eps)
            an = g[gname][ix]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(fd)), (name, ix, fd, an)


def test_extra_backward_matches_finite_differences(oracle64):
    P, W, H, E = 30, 32, 32, 3
    act_t, rs, cam = scene_inputs(P, W, H, seed=9, colmap=True, scale_mult=12.0)
    act = {k: to_np(v).astype(np.float64) for k, v in act_t.items()}
    rng = np.random.RandomState(1)
    extra = rng.randn(P, E)
    gpe = rng.randn(H * W, E)
    _, fwd = _loss_fwd(oracle64, act, rs, np.zeros((3, H, W)), np.zeros((H, W)))
    g = oracle64.extra_backward(W, H, fwd, extra, gpe)
    for _ in range(20):
        ix = (rng.randint(0, P), rng.randint(0, E))
        eps = 1e-6
        p, m = extra.copy(), extra.copy()
        p[ix] += eps
        m[ix] -= eps
        fd = ((oracle64.extra_forward(W, H, fwd, p) * gpe).sum() - (oracle64.extra_forward(W, H, fwd, m) * gpe).sum()) / (2 * eps)
        assert abs(fd - g['dL_dextra'][ix]) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize('colmap', [True, False])
def test_oracle_is_deterministic_and_threads_agree(oracle32, colmap):
    """the OpenMP oracle gives the same integers and (to rounding) the same floats whatever the thread count"""
    import os
    P, W, H = 1500, 96, 80
    act, rs, cam = scene_inputs(P, W, H, seed=2, colmap=colmap, scale_mult=3.0)
    n = to_np
    a = oracle32.rasterize_forward(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, colmap, n(rs.viewmatrix), n(rs.projmatrix),
                                   n(rs.campos), n(act['means3D']), n(act['opacity']), n(act['sh']), n(act['scales']),
                                   n(act['rotations']))
    b = oracle32.rasterize_forward(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, colmap, n(rs.viewmatrix), n(rs.projmatrix),
                                   n(rs.campos), n(act['means3D']), n(act['opacity']), n(act['sh']), n(act['scales']),
                                   n(act['rotations']))
    assert a['num_rendered'] == b['num_rendered'] > 0
    np.testing.assert_array_equal(a['color'], b['color'])
    np.testing.assert_array_equal(a['binning']['point_list'], b['binning']['point_list'])
    # sorted order inside every tile: ascending depth bits, ties by Gaussian id
    keys = a['binning']['point_list_keys']
    assert np.all(keys[1:] >= keys[:-1])
    rg = a['binning']['ranges']
    depth_bits = a['geom']['depths'].view(np.uint32)
    for t in np.nonzero(rg[:, 1] > rg[:, 0])[0][:50]:
        ids = a['binning']['point_list'][rg[t, 0]:rg[t, 1]].astype(np.int64)
        k = depth_bits[ids].astype(np.int64) * (1 << 32) + ids
        assert np.all(np.diff(k) > 0)


def test_distance_based_lbs_weightings_match_torch_autograd():
    """oracle.lbs_weights_kernel / lbs_weights_dist / knn_dist_backward (restating networks/sk_gs.py:757-770) against
    torch autograd of the reference's own expressions, fp64 on CPU"""
    import numpy as np
    import torch
    from oracle.oracle import Oracle, build
    build()
    o = Oracle('f64')
    gen = torch.Generator().manual_seed(0)
    P, M, K, dim = 300, 12, 5, 11
    pts = torch.randn(P, dim, generator=gen, dtype=torch.float64)
    jts = torch.randn(M, dim, generator=gen, dtype=torch.float64).requires_grad_(True)
    radius = torch.exp(0.3 * torch.randn(M, generator=gen, dtype=torch.float64)).requires_grad_(True)
    kweight = torch.sigmoid(torch.randn(M, generator=gen, dtype=torch.float64)).requires_grad_(True)
    g_w = torch.randn(P, K, generator=gen, dtype=torch.float64)
    d_ref, idx = o.knn_bones(pts.numpy(), jts.detach().numpy(), K)
    idx_t = torch.from_numpy(idx)
    nn_dist = (pts[:, None, :] - jts[idx_t]).square().sum(-1)
    assert np.allclose(nn_dist.detach().numpy(), d_ref, rtol=1e-12)
    # weighted kernel (sk_gs.py:759-766)
    w = torch.exp(-nn_dist / (2 * radius[idx_t] ** 2)) * kweight[idx_t] + 1e-7
    w = w / w.sum(dim=-1, keepdim=True)
    gj, gr, gk = torch.autograd.grad((w * g_w).sum(), [jts, radius, kweight])
    w_o, g = o.lbs_weights_kernel(d_ref, idx, radius.detach().numpy(), kweight.detach().numpy(), g_w.numpy())
    assert np.allclose(w_o, w.detach().numpy(), rtol=1e-12, atol=1e-15)
    assert np.allclose(g['g_radius'], gr.numpy(), rtol=1e-9, atol=1e-12) and np.allclose(g['g_weight'], gk.numpy(), rtol=1e-9, atol=1e-12)
    _, g_j = o.knn_dist_backward(pts.numpy(), jts.detach().numpy(), idx, g['g_dist'])
    assert np.allclose(g_j, gj.numpy(), rtol=1e-9, atol=1e-12)
    # dist (sk_gs.py:769-770)
    nn_dist = (pts[:, None, :] - jts[idx_t]).square().sum(-1)
    w = torch.softmax(-nn_dist / 0.7, dim=-1)
    gj, = torch.autograd.grad((w * g_w).sum(), [jts])
    w_o, g_d = o.lbs_weights_dist(d_ref, 0.7, g_w.numpy())
    assert np.allclose(w_o, w.detach().numpy(), rtol=1e-12, atol=1e-15)
    _, g_j = o.knn_dist_backward(pts.numpy(), jts.detach().numpy(), idx, g_d)
    assert np.allclose(g_j, gj.numpy(), rtol=1e-9, atol=1e-12)
