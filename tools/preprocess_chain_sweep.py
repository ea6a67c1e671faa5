#!/usr/bin/env python3
"""What is IN the dependency chain of the two per-Gaussian launches (would splitting a Gaussian over 4 lanes reach 40 %
of the HBM roof at config #1)?

A 4-lane split shortens the ARITHMETIC of a lane (the search over the joints, the SH polynomial and its gradient) and leaves the memory
round trips where they are.  This sweep removes that arithmetic instead of splitting it -- 2 joints / 1 neighbour instead of 20 / 5, SH
degree 0 instead of 3 (one coefficient: no 45-coefficient rows) -- at P = 100k (config #1: one round of resident waves) and P = 25k (less
than one workgroup per CU: the launch's time is ONE workgroup's chain).  What is left at (2 joints, degree 0) is launch + round trips +
the projection / covariance arithmetic: the floor ANY redistribution of the work over lanes keeps.
Usage: python tools/preprocess_chain_sweep.py  (GPU box)."""
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
    print(f'{"P":>8s} {"M":>3s} {"K":>2s} {"SH":>2s} {"preprocess_forward us":>22s} {"preprocess_backward us":>23s} {"skeleton fwd/bwd us":>20s}')
    for P in (100_000, 25_000):
        for (M, K) in ((20, 5), (2, 1)):
            for deg in (3, 0):
                model = SkinnedGaussians(P, M, K, sh_degree=deg, num_frames=2, seed=0, deform_net=True, learn_joints=True).to(dev)
                from sk_gs_amd.densify import sort_spatially
                sort_spatially(model)
                rs = scene.raster_settings_from_camera(scene.make_camera(W, H, seed=0), sh_degree=deg, colmap=True, device=dev)
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
                for _ in range(30):
                    step.forward_backward(rs, 0, target)
                torch.cuda.synchronize()
                prof = _C.profile_collect()
                _C.profile_enable([])
                us = lambda k: 1e3 * prof[k][0] / prof[k][1] if k in prof else float('nan')  # noqa: E731
                print(f'{P:8d} {M:3d} {K:2d} {deg:2d} {us("preprocess_forward"):22.1f} {us("preprocess_backward"):23.1f} '
                      f'{us("skeleton_forward"):9.1f} /{us("skeleton_backward"):6.1f}')
                del step, model
                torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
