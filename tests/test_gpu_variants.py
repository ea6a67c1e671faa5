"""GPU tests of the two boundary variants (INTEGRATION.md sections 2-3) and of the operator surface end to end."""
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close_robust, oracle_backward, oracle_forward, scene_inputs, to_np

pytestmark = pytest.mark.gpu


def test_variant_a_upstream_front_end(oracle32):
    """diff_gaussian_rasterization-compatible shim: wxyz quaternions, bg inside, returns (color, radii)"""
    import sk_gs_amd
    sk_gs_amd.install_as_diff_gaussian_rasterization()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    assert sys.modules['diff_gaussian_rasterization'].__name__ == 'sk_gs_amd.diff_gaussian_rasterization'
    P, W, H = 3000, 144, 112
    act, rs, cam = scene_inputs(P, W, H, seed=12, colmap=True, scale_mult=3.0, device='cuda')
    bg = torch.tensor([0.2, 0.7, 1.0], device='cuda')
    settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, bg=bg,
                                             scale_modifier=1.0, viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix,
                                             sh_degree=3, campos=rs.campos, prefiltered=False, debug=False)
    leaves = {k: act[k].clone().requires_grad_(True) for k in ('means3D', 'opacity', 'sh', 'scales')}
    rot_wxyz = act['rotations'][:, (3, 0, 1, 2)].clone().requires_grad_(True)
    means2D = torch.zeros_like(leaves['means3D'], requires_grad=True)
    color, radii = GaussianRasterizer(settings)(means3D=leaves['means3D'], means2D=means2D, opacities=leaves['opacity'],
                                                shs=leaves['sh'], scales=leaves['scales'], rotations=rot_wxyz)
    ref = oracle_forward(oracle32, act, rs)
    want = ref['color'] + (1 - ref['opacity'])[None] * to_np(bg)[:, None, None]
    assert_close_robust(color, want, 1e-4, name='color+bg')
    np.testing.assert_array_equal(to_np(radii), ref['radii'])
    g = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    color.backward(g)
    # upstream semantics: dL_dT = dot(bg, dL_dpixel)  <=>  dL_dout_opacity = -dot(bg, dL_dpixel)
    gop = -(g * bg.view(3, 1, 1)).sum(0)
    gref = oracle_backward(oracle32, ref, act, rs, g, gop)
    assert_close_robust(leaves['means3D'].grad, gref['dL_dmeans3D'], 1e-4, 1e-3, name='means3D')
    assert_close_robust(leaves['opacity'].grad, gref['dL_dopacity'], 1e-4, 1e-3, name='opacity')
    assert_close_robust(leaves['sh'].grad, gref['dL_dsh'], 1e-4, 1e-3, name='sh')
    assert_close_robust(rot_wxyz.grad[:, (1, 2, 3, 0)], gref['dL_drotations'], 1e-4, 1e-3, name='rot')
    assert_close_robust(means2D.grad, gref['dL_dmean2D'], 1e-4, 1e-3, name='means2D')


def test_variant_b_my_ext_C_registration():
    import sk_gs_amd
    from sk_gs_amd import _C
    sk_gs_amd.install_as_my_ext_C()
    try:
        assert sys.modules['my_ext._C._C'] is _C.pybind_module()      # the compiled INNER module is what is replaced
        from my_ext._C import get_C_function
        for name in _C.PYBIND_NAMES:
            assert get_C_function(name) is getattr(_C, name)
        assert get_C_function('xfm_fwd') is None
    finally:
        sk_gs_amd.uninstall_my_ext_C()


def test_pybind_names_drive_a_reference_shaped_render(oracle32):
    """What the UNMODIFIED networks/renderer/gaussian_render.py does, spelled out: every op is looked up with
    getattr(module, name) on the module install_as_my_ext_C registers as my_ext._C._C, and called with the reference's
    positional arguments (gaussian_render.py:69-93 forward, :105-108 extra forward, :132-151 extra backward chained into
    :152-174 backward, :343-347 top-k, :231 mark_visible).  Checked against the oracle."""
    from sk_gs_amd import _C
    m = _C.pybind_module()
    # the strict build (oracle operation order, shared reproducible exp): the forward is bit-exact, no branch flips to trace
    _C.set_strict_math(True)
    oracle32.set_exp_mode(1)
    try:
        P, W, H, E = 2000, 112, 80, 5  # (the scene of test_gpu_raster.py::test_other_extras_and_topk: no pixel of it sits on a
        act, rs, cam = scene_inputs(P, W, H, seed=8, colmap=True, scale_mult=3.0, device='cuda')  # branch of the extras' walk)
        g = torch.Generator().manual_seed(5)
        feat = torch.randn(P, E, generator=g).cuda()
        e = torch.Tensor([]).cuda()
        s = rs
        args = (s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.sh_degree, s.scale_modifier, s.prefiltered, s.debug,
                s.colmap, s.viewmatrix, s.projmatrix, s.campos,
                act['means3D'], act['opacity'], act['sh'], act['scales'], act['rotations'], None, e, e)
        num_rendered, color, opacity, radii, geom, binning, img, out_extra = getattr(m, 'rasterize_gaussians')(*args)
        ref = oracle_forward(oracle32, act, rs)
        assert num_rendered == ref['num_rendered']
        np.testing.assert_array_equal(to_np(radii), ref['radii'])
        np.testing.assert_array_equal(to_np(color), ref['color'])
        np.testing.assert_array_equal(to_np(opacity), ref['opacity'])
        assert out_extra is None or out_extra.numel() == 0

        pix = getattr(m, 'gaussian_rasterize_extra_forward')(W, H, num_rendered, feat, geom, binning, img)
        want = oracle32.extra_forward(W, H, ref, to_np(feat))
        assert_close_robust(pix.reshape(-1), want.reshape(-1), 1e-4, name='pybind extra fwd')

        g_pix = torch.randn(pix.shape, generator=g).cuda()
        g_extra, gm2, gconic, gop = getattr(m, 'gaussian_rasterize_extra_backward')(
            W, H, num_rendered, feat, opacity, g_pix, geom, binning, img, None, None, None)
        ge = oracle32.extra_backward(W, H, ref, to_np(feat), to_np(g_pix))
        assert_close_robust(g_extra, ge['dL_dextra'], 1e-4, 1e-3, name='pybind extra bwd')

        g_color = torch.randn(3, H, W, generator=g).cuda()
        g_opac = torch.randn(H, W, generator=g).cuda()
        keep = (gm2.clone(), gconic.clone(), gop.clone())
        out = getattr(m, 'rasterize_gaussians_backward')(
            s.scale_modifier, s.tanfovx, s.tanfovy, s.sh_degree, s.debug, s.colmap, s.viewmatrix, s.projmatrix, s.campos,
            act['means3D'], e, None, act['scales'], act['rotations'], e, act['sh'], num_rendered, radii, opacity,
            g_color, g_opac, None, gm2, gconic, gop, geom, binning, img)
        assert len(out) == 9
        gref = oracle_backward(oracle32, ref, act, rs, g_color, g_opac, grad_means2D=keep[0], grad_conic=keep[1],
                               grad_opacity=keep[2])
        names = ('dL_dmean2D', 'dL_dcolors', 'dL_dopacity', 'dL_dmeans3D', 'dL_dcov3D', 'dL_dsh', 'dL_dscales', 'dL_drotations')
        for t, nm in zip(out[:8], names):
            if nm in gref and t is not None and t.numel():
                assert_close_robust(t, gref[nm], 1e-4, 1e-3, name='pybind ' + nm)

        idx, w = getattr(m, 'gaussian_topk_weights')(3, W, H, P, num_rendered, geom, binning, img)
        ti, tw = oracle32.topk_weights(3, W, H, ref)
        assert (to_np(idx).reshape(ti.shape) != ti).mean() <= 1e-3
        assert_close_robust(w, tw, 1e-4, name='pybind topk w')
        vis = getattr(m, 'mark_visible')(act['means3D'], s.viewmatrix, s.projmatrix)
        assert vis.dtype == torch.bool and tuple(vis.shape) == (P,)
        assert getattr(m, 'simple_knn_not_there', None) is None
    finally:
        _C.set_strict_math(False)
        oracle32.set_exp_mode(0)


@pytest.mark.parametrize('B,D,deg', [(20, 3, 10), (20, 1, 6), (512, 3, 10), (512, 8, 4), (1, 3, 0), (4097, 2, 5)])
def test_freq_encode_under_its_pybind_name(B, D, deg):
    """freq_encode_forward / _backward with the reference's positional arguments (freqencoder.cu:66-105; called from
    networks/encoders/freq_encoder.py:33,53) against the numpy restatement of its two kernels (oracle/oracle.py).  The
    sizes are the ones the path uses: 20 bones or 512 superpoints x (3-d position, degree 10 | 1-d time, degree 6)."""
    from oracle import oracle as om
    from sk_gs_amd import _C
    m = _C.pybind_module()
    Cn = D + 2 * D * deg
    g = torch.Generator().manual_seed(B * 131 + D * 7 + deg)
    x = (torch.rand(B, D, generator=g) * 2 - 1).cuda()
    out = torch.full((B, Cn), float('nan'), device='cuda')
    assert getattr(m, 'freq_encode_forward')(x, B, D, deg, Cn, out) is None
    want = om.freq_encode_forward(to_np(x), deg)
    # |argument| reaches 2^9: one fp32 ulp of the argument is 6e-5 there, sin is 1-Lipschitz
    np.testing.assert_allclose(to_np(out), want, rtol=0, atol=1e-4 if deg > 6 else 2e-6)
    np.testing.assert_array_equal(to_np(out)[:, :D], to_np(x))
    grad = torch.randn(B, Cn, generator=g).cuda()
    gx = torch.full((B, D), float('nan'), device='cuda')    # written, not accumulated (freqencoder.cu:59)
    getattr(m, 'freq_encode_backward')(grad, out, B, D, deg, Cn, gx)
    gwant = om.freq_encode_backward(to_np(grad), to_np(out), D, deg)
    scale = max(np.abs(gwant).max(), 1e-30)
    assert np.abs(to_np(gx) - gwant).max() / scale < 1e-5
    # the analytic derivative: d/dx sin(2^f x) = 2^f cos(2^f x) -- the same numbers, from torch autograd
    xt = x.clone().requires_grad_(True)
    from sk_gs_amd.deform_net import freq_encode_torch
    (freq_encode_torch(xt, deg) * grad).sum().backward()
    assert float((xt.grad - gx).abs().max()) / scale < 1e-4
    with pytest.raises(_C.SkgsError):
        m.freq_encode_forward(x, B, D, deg, Cn + 1, out)
    with pytest.raises(_C.SkgsError):
        m.freq_encode_forward(x.cpu(), B, D, deg, Cn, out)


def test_render_dict_other_extras_and_detach(oracle32):
    """render(**net_out, raster_settings, name=extra) keys, other-extras chaining and detach_other_extra"""
    from sk_gs_amd.renderer.gaussian_render import render, topk_weights
    P, W, H = 2500, 96, 96
    act, rs, cam = scene_inputs(P, W, H, seed=21, colmap=True, scale_mult=3.0, device='cuda')
    feat = torch.randn(P, 7, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
    pts = act['means3D'].clone().requires_grad_(True)
    out = render(pts, act['opacity'], rs, scales=act['scales'], rotations=act['rotations'], sh_features=act['sh'],
                 feat=feat)
    for k in ('images', 'opacity', 'viewspace_points', 'visibility_filter', 'radii', 'extras', 'buffer', 'feat'):
        assert k in out
    assert out['extras'] is None and tuple(out['feat'].shape) == (W, H, 7)
    ref = oracle_forward(oracle32, act, rs)
    assert bool((out['visibility_filter'].cpu().numpy() == (ref['radii'] > 0)).all())
    gf = torch.randn(W, H, 7, generator=torch.Generator().manual_seed(3)).cuda()
    (out['feat'] * gf).sum().backward()
    ge = oracle32.extra_backward(W, H, ref, to_np(feat), to_np(gf))
    assert_close_robust(feat.grad, ge['dL_dextra'], 1e-4, 1e-3, name='feat grad')
    # gradients of the extra pass reach the geometry through means2D / conic / opacity (chained into the main backward)
    assert float(pts.grad.abs().max()) > 0
    rs_detached = rs._replace(detach_other_extra=True)
    pts2 = act['means3D'].clone().requires_grad_(True)
    out2 = render(pts2, act['opacity'], rs_detached, scales=act['scales'], rotations=act['rotations'],
                  sh_features=act['sh'], feat=feat.detach().requires_grad_(True))
    (out2['feat'] * gf).sum().backward()
    assert float(pts2.grad.abs().max()) == 0.0
    idx, w = topk_weights(2, out['buffer'])
    assert tuple(idx.shape) == (H, W, 2) and tuple(w.shape) == (H, W, 2)


@pytest.mark.parametrize('P', [2000, 3, 1, 50_001])
def test_simple_knn_under_its_pybind_name(P):
    """``simple_knn`` (my_ext/_C/src/other/knn.cu:192-205; what create_from_pcd resolves, gaussian_splatting.py:211-213): the mean
    squared distance to the three nearest OTHER points, against a brute-force restatement (exact three smallest of the same
    fp32 distances; fewer than four points: missing neighbours count as FLT_MAX, as upstream)"""
    from sk_gs_amd import _C
    m = _C.pybind_module()
    g = torch.Generator().manual_seed(P)
    pts = (torch.rand(P, 3, generator=g) * 2.6 - 1.3).cuda()
    got = getattr(m, 'simple_knn')(pts)
    assert tuple(got.shape) == (P,) and got.dtype == torch.float32
    n = min(P, 4000)  # (the check is O(n P) on the host)
    d = ((pts[:n, None, :] - pts[None, :, :]) ** 2).sum(-1)
    d[torch.arange(n), torch.arange(n)] = float('inf')
    best = torch.sort(d, dim=1).values[:, :3]
    best = torch.where(torch.isinf(best), torch.full_like(best, 3.402823466e+38), best)
    if best.shape[1] < 3:
        best = torch.cat([best, torch.full((n, 3 - best.shape[1]), 3.402823466e+38, device='cuda')], 1)
    want = (best[:, 0] + best[:, 1] + best[:, 2]) / 3.0
    finite = torch.isfinite(want) & (want < 1e30)
    assert rel_err_t(got[:n][finite], want[finite]) <= 1e-5
    with pytest.raises(_C.SkgsError):
        m.simple_knn(pts.cpu())


def rel_err_t(a, b):
    return float(((a - b).abs() / b.abs().clamp_min(1e-30)).max()) if b.numel() else 0.0
