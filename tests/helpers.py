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


OBSERVED = []  # what the comparisons actually saw (conftest dumps it and checks it against the committed baseline)

# The gate of the robust comparison.  Round 2 allowed 1e-4 of the pixels / 1e-3 of the gradient elements to exceed the
# tolerance by up to 3e-2 -- 300x more room than any run had used, which hid a 10x growth of the worst case.  Now: at
# most 2e-5 of the elements (or ONE element of a small tensor) over the tolerance, none over 2e-3 (~3x the worst ever
# observed, 7.1e-4), whatever the caller asks for; and tests/conftest.py fails the session when any recorded worst case
# grows more than 2x over tests/golden/parity_observed_baseline.json.
MAX_OUTLIER_FRAC = 2e-5
MAX_HARD = 2e-3
# A pixel the census has TRACED to a branch flip may be off by what one contributor at the alpha cut is worth: alpha T c <= c / 255
# for the contributor itself plus the factor (1 - alpha) on everything behind it, another <= 1 / 255 of the pixel -- 2 / 255 of the
# image's scale (a 400-scene sweep of tests/test_gpu_fuzz.py's generator found one such pixel at 2.2e-3, 5e-8 from its branch).
FLIP_HARD = 2.0 / 255.0
# FlipCensus.check_rows, second branch: a row more than 1e-4 from the fp32 oracle must be within max(1e-4, REF_ERR_FACTOR x ref_err) OF
# THE TRUTH, ref_err = the reference arithmetic's own distance from the fp64 oracle on that tensor (see there; tests/golden/
# fuzz_bounds.json is the table of every tensor that needs the branch)
REF_ERR_FACTOR = 2.0


def _record(name, d, tol, frac, allowed, hard, **kw):
    import os
    test = os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]
    OBSERVED.append(dict(test=test, name=name, elements=int(d.size), tol=tol, frac_over_tol=frac, allowed_frac=allowed,
                         max_err=float(d.max()), hard=hard, **kw))


def assert_close_robust(a, b, tol=1e-4, outlier_frac=MAX_OUTLIER_FRAC, hard=MAX_HARD, name=''):
    """max-norm-relative comparison that tolerates threshold flips.

    The blend uses the hardware exp; a (pixel, Gaussian) pair whose alpha sits within an ulp of the 1/255 cut, or
    whose transmittance sits within an ulp of the 1e-4 stop, may take the other branch than the oracle (any two
    implementations with different exp rounding do that, the CUDA reference included).  Such flips are rare and
    bounded: at most `outlier_frac` (<= MAX_OUTLIER_FRAC; one element of a small tensor) of the elements may exceed
    `tol`, none may exceed `hard` (<= MAX_HARD).  Where the oracle is at hand use FlipCensus instead: it demands that
    every element over `tol` is TRACED to a flip."""
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64).reshape(to_np(a).shape)
    if b.size == 0:
        return
    outlier_frac, hard = min(outlier_frac, MAX_OUTLIER_FRAC), min(hard, MAX_HARD)
    allowed = max(outlier_frac, 1.5 / b.size)
    scale = max(np.abs(b).max(), 1e-30)
    d = np.abs(a - b) / scale
    frac = float((d > tol).mean())
    _record(name, d, tol, frac, allowed, hard)
    print(f'[parity] {name}: {frac:.3e} of {d.size} elements over {tol:g} (allowed {allowed:.1e}), max {d.max():.3e}')
    assert frac <= allowed, f'{name}: {frac:.2e} of the elements exceed {tol} (max {d.max():.2e})'
    assert d.max() <= hard, f'{name}: max rel err {d.max():.2e} > {hard}'


class FlipCensus:
    """Parity against the oracle with every out-of-tolerance element TRACED to a branch flip.

    The reference walk (gaussian_render.cu:78-100) has three data-dependent branches per (pixel, Gaussian) pair: power > 0,
    alpha < 1/255, T(1-alpha) < 1e-4.  Two implementations that round exp / the quadratic form differently take different
    branches at pairs that sit within rounding distance of one.  Two sources say where:

    * `census` (exact): the fingerprint of the list entries each pixel blended, from the implementation under test
      (HIP: `_C.render_census`, same kernel source, image compared bit for bit with the product kernel's) against the
      oracle's `render_census`.  Pixels whose fingerprints differ are the flipped pixels -- counted, and each must sit
      within `eps` (in units of the local rounding error, `oracle.render_margins`) of a branch: a flip further away
      would be a bug, not a rounding.  Observed on the MI355X at the four BASELINE sizes (profiles/r03_a_flip_census_ab.txt):
      0 / 2 / 5 / 3 flipped pixels, every one within 7e-8 of its branch (one ulp); the default eps is 7x that.
    * without a census: every pixel within `eps` of a branch counts as possibly flipped.

    A pixel may differ from the oracle by more than `tol` only if it flipped; a per-Gaussian gradient row may only if
    the Gaussian (nearly) contributes to a flipped pixel.  Everything else is held to `tol` with no allowance, traced
    elements to `hard`."""

    def __init__(self, oracle, ref, W, H, eps=5e-7, tol=1e-4, hard=MAX_HARD, name=''):
        self.o, self.ref, self.W, self.H = oracle, ref, W, H
        self.eps, self.tol, self.hard, self.name = eps, tol, hard, name
        self.margin = oracle.render_margins(W, H, ref)
        self.flipped = None
        self.rows = None

    def check_image(self, color, opacity, census=None):
        """color [3,H,W], opacity [H,W] (and the census fingerprint [H,W,2]) of the implementation under test"""
        near = self.margin < self.eps
        extra = dict(near_branch_pixels=int(near.sum()))
        if census is not None:
            mine = to_np(census).astype(np.int64).reshape(self.H, self.W, 2) & 0xffffffff
            theirs = self.o.render_census(self.W, self.H, self.ref).astype(np.int64)
            flipped = (mine != theirs).any(-1)
            worst = float(self.margin[flipped].max()) if flipped.any() else 0.0
            extra.update(flipped_pixels=int(flipped.sum()), flipped_max_margin=worst)
            print(f'[census] {self.name}: {int(flipped.sum())} of {flipped.size} pixels took a different branch than the oracle; '
                  f'largest margin among them {worst:.2e} (rounding units; {int(near.sum())} pixels are within {self.eps:g})')
            assert not (flipped & ~near).any(), \
                f'{self.name}: {int((flipped & ~near).sum())} pixels flipped a branch further than {self.eps} from it ({worst:.2e})'
        else:
            flipped = near
        self.flipped = flipped
        for nm, got, want in (('color', color, self.ref['color']), ('opacity', opacity, self.ref['opacity'])):
            got, want = to_np(got).astype(np.float64), np.asarray(want, np.float64).reshape(to_np(got).shape)
            d = np.abs(got - want) / max(np.abs(want).max(), 1e-30)
            dp = d.max(0) if d.ndim == 3 else d
            over = dp > self.tol
            clean = dp[~flipped]
            _record(f'{self.name} {nm}', d, self.tol, float((d > self.tol).mean()), 0.0, self.hard,
                    traced_pixels=int((over & flipped).sum()), untraced_max=float(clean.max()), **extra)
            print(f'[census] {self.name} {nm}: {int(over.sum())} pixels over {self.tol:g}; max over the pixels that did not flip '
                  f'{clean.max():.3e}; max {d.max():.3e}')
            assert not (over & ~flipped).any(), \
                f'{self.name} {nm}: {int((over & ~flipped).sum())} pixels over {self.tol} that no branch flip explains ' \
                f'(max {clean.max():.2e})'
            assert d.max() <= max(self.hard, FLIP_HARD), f'{self.name} {nm}: max rel err {d.max():.2e} > {max(self.hard, FLIP_HARD):.2e}'
        self.rows = self.o.render_touching(self.W, self.H, self.ref, flipped)
        return int(flipped.sum())

    def check_rows(self, got, want, name, exact=None):
        """per-Gaussian tensor [P, ...]: rows of Gaussians that touch a flipped pixel <= hard, all others <= tol.

        ``exact`` (the same tensor from the fp64 oracle) tells how far the REFERENCE arithmetic itself is from the true value on this
        tensor: ``ref_err = max |fp32 oracle - fp64 oracle| / scale``.  Some scenes (a camera inside the cloud, splats covering the image)
        are ill-conditioned in fp32: there the fp32 oracle is off by 1e-4 ... 0.45 of the tensor's scale, the strict build reproduces it
        to 1e-6, and the product build -- another summation order of the same terms -- lands elsewhere inside the same error ball.  An
        untraced row passes when it is within ``tol`` (1e-4) of the fp32 oracle -- the north star's literal bar -- OR when it is as close
        to the TRUE value as the reference arithmetic is:  |product - fp64 oracle| <= max(tol, REF_ERR_FACTOR x ref_err).  (Round 5
        widened the bar AROUND THE FP32 ORACLE by 2.5 x ref_err, which let the product sit 3.5 x ref_err from the truth.)  Why the factor
        is 2 and not 1: the product and the fp32 oracle are two fp32 evaluations of the same ill-conditioned sums in different orders --
        two draws from one error distribution; over the 538 tensors of the 72 scenes of tests/test_gpu_fuzz.py the product's distance to the
        truth is 0.0 ... 1.74 x the oracle's (median 1.00: on 530 of 538 tensors the product follows the fp32 oracle to 1e-5 and simply
        shares its error).  tests/golden/fuzz_bounds.json lists every tensor that needs the second branch (8 of 538) with ref_err, both
        errors and its bound; tests/test_host_cpu.py checks the table against REF_ERR_FACTOR."""
        assert self.rows is not None, 'check_image first'
        got = to_np(got).astype(np.float64)
        want = np.asarray(want, np.float64).reshape(got.shape)
        if want.size == 0:
            return
        scale = max(np.abs(want).max(), 1e-30)
        d = np.abs(got - want) / scale
        dr = d.reshape(d.shape[0], -1).max(1)
        over = dr > self.tol
        extra = {}
        bound = self.tol
        if exact is not None:
            ex = np.asarray(exact, np.float64).reshape(got.shape)
            ref_err = float(np.abs(want - ex).max() / scale)
            d64 = (np.abs(got - ex) / scale).reshape(d.shape[0], -1).max(1)
            clean64 = d64[~self.rows]
            extra = dict(ref_err=ref_err, untraced_max_vs_fp64=float(clean64.max()) if clean64.size else 0.0)
            bound = max(self.tol, REF_ERR_FACTOR * ref_err)
            extra['bound_vs_fp64'] = bound
            if over.any():
                over = over & (d64 > bound)          # within tol of the fp32 oracle, OR as close to the truth as the reference is
                print(f'[census] {self.name} {name}: the fp32 oracle is {ref_err:.2e} from the fp64 one: rows over {self.tol:g} of it are '
                      f'held to {bound:.2e} of the fp64 oracle (worst untraced: {extra["untraced_max_vs_fp64"]:.2e})')
        clean = dr[~self.rows]
        _record(f'{self.name} {name}', d, self.tol, float((d > self.tol).mean()), 0.0, self.hard,
                traced_rows=int((over & self.rows).sum()), rows_touching_a_flip=int(self.rows.sum()),
                untraced_max=float(clean.max()) if clean.size else 0.0, **extra)
        print(f'[census] {self.name} {name}: {int((dr > self.tol).sum())} rows over {self.tol:g} ({int(self.rows.sum())} of {dr.size} rows '
              f'touch a flipped pixel); max over the others {clean.max() if clean.size else 0.0:.3e}; max {d.max():.3e}')
        assert not (over & ~self.rows).any(), \
            f'{self.name} {name}: {int((over & ~self.rows).sum())} rows over {self.tol:.0e} of the fp32 oracle' + \
            (f' and over {bound:.2e} of the fp64 one' if bound != self.tol else '') + \
            f' that touch no flipped pixel (max {clean.max():.2e})'
        assert d.max() <= max(self.hard, bound + extra.get('ref_err', 0.0)), \
            f'{self.name} {name}: max rel err {d.max():.2e} > {max(self.hard, bound + extra.get("ref_err", 0.0)):.2e}'


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


def sp_net_relu_masks_agree(net, run, x, t):
    """True when the kernel's saved activations and the torch body took the same side of every ReLU.  A pre-activation within rounding
    of 0 may land on either side in two fp32 evaluations; ONE such flip moves a weight gradient by a row's whole contribution (~1e-2
    of its largest element), which says nothing about either evaluation"""
    from sk_gs_amd.deform_net import freq_encode_torch
    M = x.shape[0]
    Mp = (M + 15) // 16 * 16
    saved = run.saved.view(torch.float32)
    with torch.no_grad():
        t_emb = freq_encode_torch(t.view(-1, 1), net.t_degree).expand(M, net.t_dim)
        if net.is_blender:
            t_emb = net.timenet(t_emb)
        x_emb = freq_encode_torch(x, net.p_degree)
        h = torch.cat([x_emb, t_emb], -1)
        for i, layer in enumerate(net.linear):
            h = torch.relu(layer(h))
            Y = saved[Mp * 96 + i * Mp * 256: Mp * 96 + (i + 1) * Mp * 256].view(Mp, 256)[:M]
            if bool(((Y > 0) != (h > 0)).any()):
                return False
            if i in net.skips:
                h = torch.cat([x_emb, t_emb, h], -1)
    return True
