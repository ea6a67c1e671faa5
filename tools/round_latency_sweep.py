#!/usr/bin/env python3
"""Are the per-Gaussian launches at config #1 bounded by HBM or by ONE ROUND of resident waves?

The step of bench.py (FusedViewStep, eager, HIP-event timing per kernel) at 800 x 800 with P = 25k ... 800k Gaussians whose
scales shrink with P^(-1/3), so the image-space work per Gaussian stays alike.  A kernel that streams is
proportional to P; a kernel that is one round of resident waves (P / 256 workgroups on 256 CUs x 6 slots) does not get faster
when P shrinks below that round: its time is the length of a lane's dependency chain.  Prints us per launch, us per 100k Gaussians
and the fraction of the 8 TB/s roof on SURVEY 8(d)'s bytes.
Usage: python tools/round_latency_sweep.py  (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sk_gs_amd import _C, scene  # noqa: E402
from sk_gs_amd.fused_step import FusedViewStep  # noqa: E402
from sk_gs_amd.model import SkinnedGaussians  # noqa: E402


def main():
    dev = torch.device('cuda')
    W = H = 800
    M, K = 20, 5
    rows = []
    for P in (25_000, 50_000, 100_000, 200_000, 400_000, 800_000):
        model = SkinnedGaussians(P, M, K, sh_degree=3, num_frames=2, seed=0, deform_net=True, learn_joints=True).to(dev)
        from sk_gs_amd.densify import sort_spatially
        sort_spatially(model)
        rs = scene.raster_settings_from_camera(scene.make_camera(W, H, seed=0), sh_degree=3, colmap=True, device=dev)
        target = torch.rand(3, H, W, device=dev)
        _C.config.sync_num_rendered = True
        with torch.no_grad():
            buf = model.render(rs, time_id=0, background=torch.ones(3, device=dev))['buffer']
        R, longest = buf.R, _C.read_status(buf.geomBuffer)['max_tile_count']
        _C.config.sync_num_rendered = False
        for p in model.parameters():
            p.grad = None
        bucket = ((int(longest * 1.5) + 63) // 64) * 64
        step = FusedViewStep(model, W, H, capacity=int(R * 2) + 4096, background=torch.ones(3, device=dev), tile_bucket=bucket)
        for _ in range(5):
            step.forward_backward(rs, 0, target)
        torch.cuda.synchronize()
        _C.profile_enable(None)
        n = 20
        for _ in range(n):
            step.forward_backward(rs, 0, target)
        torch.cuda.synchronize()
        prof = _C.profile_collect()
        _C.profile_enable([])
        assert step.status()['overflow_events'] == 0
        alg = {'preprocess_forward': P * (355 + 4 * M + 12 * K), 'preprocess_backward': P * (627 + 40 + 4 * K - 44), 'scatter': 28 * P + 12 * R,
               'tile_sort': 16 * R, 'render_forward': 40 * R + 20 * W * H, 'render_backward': 40 * R + 24 * W * H + 44 * P}
        rows.append((P, R, {k: (1e3 * prof[k][0] / prof[k][1], alg[k]) for k in alg if k in prof}))
        del step, model
        torch.cuda.empty_cache()
    names = ['preprocess_forward', 'preprocess_backward', 'scatter', 'tile_sort', 'render_forward', 'render_backward']
    print(f'{"P":>8s} {"R":>9s} ' + ' '.join(f'{n[:19]:>26s}' for n in names))
    print(f'{"":>8s} {"":>9s} ' + ' '.join(f'{"us  us/100k  frac":>26s}' for _ in names))
    for P, R, d in rows:
        cells = []
        for n in names:
            if n in d:
                us, b = d[n]
                cells.append(f'{us:8.1f} {us * 1e5 / P:8.1f} {b / (us * 1e-6) / 8e12:8.3f}')
            else:
                cells.append(' ' * 26)
        print(f'{P:8d} {R:9d} ' + ' '.join(cells))


if __name__ == '__main__':
    main()
