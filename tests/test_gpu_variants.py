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
    sk_gs_amd.install_as_my_ext_C()
    from my_ext._C import get_C_function
    for name in ('rasterize_gaussians', 'rasterize_gaussians_backward', 'gaussian_rasterize_extra_forward',
                 'gaussian_rasterize_extra_backward', 'gaussian_topk_weights', 'mark_visible'):
        assert callable(get_C_function(name))


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
