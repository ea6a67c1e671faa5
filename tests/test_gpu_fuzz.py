"""Seeded sweep of odd rasterizer configurations against the oracle: image sizes that are no multiple of the 16-pixel tile or
of the 8-pixel quadrant, one to a few thousand Gaussians, cameras from inside the cloud (Gaussians straddling the near
plane, splats covering the whole image) to far outside it (sub-pixel splats), every SH degree, both matrix conventions,
extras, precomputed colours.  Every case runs the full forward comparison of tests/test_gpu_raster.py::check_forward
(radii, records, tile ranges and sorted lists bit-exact; image through the exact flip census) and the backward against the
oracle with every element over the tolerance traced to a flipped pixel.  The cases are fixed by their seeds: a failure
names a reproducible scene."""
import math

import numpy as np
import pytest
import torch

from helpers import assert_close_robust, oracle_backward, oracle_forward
from test_gpu_raster import GRAD_NAMES, TOL, check_forward, hip_backward

pytestmark = pytest.mark.gpu


def _case(seed):
    """one random scene: everything drawn from `seed`"""
    from sk_gs_amd import scene
    r = np.random.RandomState(7000 + seed)
    W, H = int(r.randint(17, 300)), int(r.randint(17, 260))
    P = int(r.choice([1, 2, 7, 63, 64, 65, 500, 2000, 4000]))
    colmap = bool(r.randint(2))
    degree = int(r.randint(4))
    scale_mult = float(r.choice([0.05, 0.5, 1.0, 3.0, 8.0, 25.0])) * (2000.0 / max(P, 50)) ** (1. / 3.) * 0.4
    radius = float(r.choice([0.3, 1.0, 2.5, 4.0, 12.0]))  # 0.3 / 1.0: the camera sits inside the cloud of +-1.3
    fovx = float(r.choice([0.3, 0.6911, 1.4]))
    E = int(r.choice([0, 0, 0, 1, 3]))
    use_colors = bool(r.randint(4) == 0)
    g = scene.make_gaussians(P, seed=seed, sh_degree=3, scale_mult=scale_mult)
    act = scene.activate(g)
    cam = scene.make_camera(W, H, seed=seed, radius=radius, fovx=fovx, near=0.05 if radius < 2 else 2.0, far=radius + 3.0)
    rs = scene.raster_settings_from_camera(cam, sh_degree=degree, colmap=colmap, device='cuda')
    act = {k: v.cuda() for k, v in act.items()}
    tg = torch.Generator().manual_seed(seed)
    extras = torch.randn(P, E, generator=tg).cuda() if E else None
    colors = torch.rand(P, 3, generator=tg).cuda() if use_colors else None
    return dict(W=W, H=H, P=P, colmap=colmap, degree=degree, E=E, radius=radius, fovx=fovx, scale_mult=round(scale_mult, 3),
                colors=use_colors), act, rs, extras, colors


# beyond the first 64: the scenes a 1000-seed sweep singled out -- 240: a flipped pixel worth 2.2e-3; 762: a flipped pixel under
# extras; 394, 474, 570, 640, 753, 1008: ill-conditioned in fp32 (the fp32 oracle itself is 1e-4 ... 0.45 from the fp64 one)
@pytest.mark.parametrize('seed', list(range(64)) + [240, 394, 474, 570, 640, 753, 762, 1008])
def test_random_scene_against_the_oracle(oracle32, oracle64, seed):
    desc, act, rs, extras, colors = _case(seed)
    W, H, P, E = desc['W'], desc['H'], desc['P'], desc['E']
    ref, fwd = check_forward(oracle32, act, rs, extras=extras, colors=colors)
    R = int(ref['num_rendered'])
    tg = torch.Generator().manual_seed(100 + seed)
    dL_dcolor = torch.randn(3, H, W, generator=tg).cuda()
    dL_dopacity = torch.randn(H, W, generator=tg).cuda()
    dL_dextra = torch.randn(E, H, W, generator=tg).cuda() if E else None
    gref = oracle_backward(oracle32, ref, act, rs, dL_dcolor, dL_dopacity, extras, dL_dextra, colors=colors)
    # the same gradients in fp64: how far the reference arithmetic itself is from the true value on this scene
    g64 = oracle_backward(oracle64, oracle_forward(oracle64, act, rs, extras, colors), act, rs, dL_dcolor, dL_dopacity, extras,
                          dL_dextra, colors=colors)
    got = hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity, extras, dL_dextra, colors=colors)
    for name, t in zip(GRAD_NAMES, got[:8]):
        assert bool(torch.isfinite(t).all()), (desc, name)
        if name == 'dL_dsh' and colors is not None:
            continue  # (no SH input: an empty gradient)
        want = gref[name]
        if name == 'dL_dcolors' and colors is None:
            continue  # (colours come from the SH: the oracle reports their gradient under dL_dsh)
        ref['census'].check_rows(t, want, f'{name} {desc}', exact=g64[name])
    if E:
        ref['census'].check_rows(got[8], gref['dL_dextras'], f'dL_dextras {desc}', exact=g64['dL_dextras'])
    else:
        assert got[8] is None
    # nothing rendered: every gradient is exactly zero
    if R == 0:
        for t in got[:8]:
            assert float(t.abs().max()) == 0.0 if t.numel() else True
