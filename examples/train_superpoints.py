#!/usr/bin/env python3
"""Minimal training loop of the SUPERPOINT stage (the reference's stage `sp`, networks/sk_gs.py:830-856) on a synthetic scene.

    SuperpointGaussians       : the Gaussians, their 8 hyper coordinates, 512 superpoints with `sp_deform_net` (8 x 256) and the
                                weighting of calc_LBS_weight -- `--lbs-method W` is the reference's default (exps/default.yaml:35:
                                a dense [P, M] logit table), `weighted_kernel` the class default
    FusedSuperpointStep       : sp_deform_net -> 3 + 8-d search + weighting -> skinning -> rasterize -> 0.8 L1 + 0.2 (1 - SSIM)
                                -> backward, direct calls into libskgs_hip.so, gradients written into the parameters' .grad
    FusedSuperpointTrainStep  : + FusedAdam (the per-Gaussian rows' update on the idle CUs of the network's backward launches,
                                `W`: the logit table updated sparsely) -- ONE captured hipGraph serves all views
    densify.sort_spatially    : the Gaussians along a Z-order curve, once, before the step's buffers and graph exist

    python examples/train_superpoints.py [--gaussians 20000 --superpoints 512 --size 256 --iters 300 --lbs-method W]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gaussians', type=int, default=20000)
    ap.add_argument('--superpoints', type=int, default=512)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--lr', type=float, default=2e-4)
    ap.add_argument('--lbs-method', choices=('weighted_kernel', 'kernel', 'dist', 'W'), default='weighted_kernel')
    args = ap.parse_args()

    from sk_gs_amd import _C, scene
    from sk_gs_amd.densify import sort_spatially
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.superpoint import FusedSuperpointStep, FusedSuperpointTrainStep, SuperpointGaussians
    from sk_gs_amd.train_step import GraphedSteps
    from sk_gs_amd.view_slot import ViewTable

    assert torch.cuda.is_available(), 'needs a GPU (the product path has no CPU fallback)'
    dev = torch.device('cuda', 0)
    P, M, S, V = args.gaussians, args.superpoints, args.size, args.views
    # "ground truth": the same scene with other colours, a moved network and other hyper coordinates
    teacher = SuperpointGaussians(P, M, 5, num_frames=V, seed=1, lbs_method=args.lbs_method).to(dev)
    model = SuperpointGaussians(P, M, 5, num_frames=V, seed=1, lbs_method=args.lbs_method).to(dev)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for net in (teacher.sp_deform_net,):
            net.gaussian_warp.weight.mul_(30.0)
            net.gaussian_rotation.weight.mul_(30.0)
        model._features_dc.add_(0.3 * torch.randn(model._features_dc.shape, generator=g).to(dev))
    cams = [scene.make_camera(S, S, seed=i) for i in range(V)]
    settings = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
    background = torch.ones(3, device=dev)
    _C.config.sync_num_rendered = True
    targets, R_max, longest = [], 0, 0
    with torch.no_grad():
        for v in range(V):
            out = teacher.render(settings[v], time_id=v, background=background)
            targets.append(out['images'].clamp(0, 1).contiguous())
            for o in (out, model.render(settings[v], time_id=v, background=background)):  # tile-list sizes: both ends of the fit
                R_max = max(R_max, o['buffer'].R)
                longest = max(longest, _C.read_status(o['buffer'].geomBuffer)['max_tile_count'])
    _C.config.sync_num_rendered = False
    del teacher
    sort_spatially(model)
    table = ViewTable(settings, [float(model.frame_times[v]) for v in range(V)], list(range(V)), torch.stack(targets), dev)
    # fixed slots per tile: 3 x the longest list seen (a list that outgrows them is counted, not rendered: `status()` below;
    # a long run wraps the step in sk_gs_amd.overflow.OverflowGuard as examples/train_views.py does)
    step = FusedSuperpointStep(model, S, S, capacity=int(R_max * 3) + 4096, background=background,
                               tile_bucket=((int(longest * 3) + 63) // 64) * 64, view_table=table)
    opt = FusedAdam(model.param_groups(lr=args.lr), eps=1e-15)
    train = FusedSuperpointTrainStep(step, opt)
    table.set_order(list(range(V)))
    graph = GraphedSteps(lambda _: train(), collect_garbage=False)
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.iters):
        graph(0)
        if it % 25 == 0 or it == args.iters - 1:
            losses.append(float(step.loss3[0]))
            print(f'iter {it:5d}  loss {losses[-1]:.5f}', flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = step.status()
    assert st['overflow_events'] == 0 and st['pairs_overflow_events'] == 0, st
    assert all(l == l for l in losses), 'NaN loss'
    print(f'{args.iters} iterations in {dt:.2f} s ({args.iters / dt:.0f} it/s), loss {losses[0]:.5f} -> {losses[-1]:.5f}, '
          f'stage sp, LBS_method {args.lbs_method}, sparse logit update: {step.sparse_logits}')
    return losses


if __name__ == '__main__':
    main()
