"""Shared helpers of the parity tests (comparison metrics follow the reference's get_rel_error,
my_ext/utils/test_utils.py:6-21: max|a-b| / max|b|)."""
import numpy as np
import torch


def to_np(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def rel_err(a, b) -> float:
    """max-norm relative error, the reference's metric"""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    denom = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    return float(np.abs(a - b).max() / denom) if b.size else 0.0


def frac_outliers(a, b, rtol, atol) -> float:
    """fraction of elements with |a-b| > atol + rtol*|b|"""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    if b.size == 0:
        return 0.0
    return float((np.abs(a - b) > atol + rtol * np.abs(b)).mean())


OBSERVED = []  # (name, elements, fraction over tol, tol, max error): what the comparisons actually saw (conftest dumps it)


def assert_close_robust(a, b, tol=1e-4, outlier_frac=1e-4, hard=3e-2, name=''):
    """max-norm-relative comparison that tolerates threshold flips.

    The blend uses the hardware exp; a (pixel, Gaussian) pair whose alpha sits within an ulp of the 1/255 cut, or
    whose transmittance sits within an ulp of the 1e-4 stop, may take the other branch than the oracle (any two
    implementations with different exp rounding do that, the CUDA reference included).  Such flips are rare and
    bounded: at most `outlier_frac` of the elements may exceed `tol`, none may exceed `hard`."""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64).reshape(to_np(a).shape)
    if b.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    d = np.abs(a - b) / scale
    frac = float((d > tol).mean())
    OBSERVED.append(dict(name=name, elements=int(d.size), tol=tol, frac_over_tol=frac, allowed_frac=outlier_frac,
                         max_err=float(d.max()), hard=hard))
    print(f'[parity] {name}: {frac:.3e} of {d.size} elements over {tol:g} (allowed {outlier_frac:g}), max {d.max():.3e}')
    assert frac <= outlier_frac, f'{name}: {frac:.2e} of the elements exceed {tol} (max {d.max():.2e})'
    assert d.max() <= hard, f'{name}: max rel err {d.max():.2e} > {hard}'


def scene_inputs(P, W, H, seed=0, colmap=True, sh_degree=3, scale_mult=1.0, device='cpu'):
    from sk_gs_amd import scene
    g = scene.make_gaussians(P, seed=seed, sh_degree=3, scale_mult=scale_mult)
    act = scene.activate(g)
    cam = scene.make_camera(W, H, seed=seed)
    rs = scene.raster_settings_from_camera(cam, sh_degree=sh_degree, colmap=colmap, device=device)
    act = {k: v.to(device) for k, v in act.items()}
    return act, rs, cam


def oracle_forward(o, act, rs, extras=None, colors=None, cov3D=None):
    n = to_np
    use_sh = colors is None
    return o.rasterize_forward(
        rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.scale_modifier, rs.colmap,
        n(rs.viewmatrix), n(rs.projmatrix), n(rs.campos), n(act['means3D']), n(act['opacity']),
        n(act['sh']) if use_sh else None, n(act['scales']) if cov3D is None else None,
        n(act['rotations']) if cov3D is None else None, n(extras), n(colors), n(cov3D))


def oracle_backward(o, fwd, act, rs, dL_dcolor, dL_dopacity, extras=None, dL_dextra=None, colors=None, cov3D=None,
                    grad_means2D=None, grad_conic=None, grad_opacity=None):
    n = to_np
    use_sh = colors is None
    return o.rasterize_backward(
        fwd, rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.scale_modifier, rs.colmap,
        n(rs.viewmatrix), n(rs.projmatrix), n(rs.campos), n(act['means3D']), n(act['sh']) if use_sh else None,
        n(act['scales']) if cov3D is None else None, n(act['rotations']) if cov3D is None else None, n(dL_dcolor),
        n(dL_dopacity), n(extras), n(dL_dextra), n(colors), n(cov3D), n(grad_means2D), n(grad_conic), n(grad_opacity))
