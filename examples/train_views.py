#!/usr/bin/env python3
"""Minimal training loop on a synthetic skinned scene: what a user of the reference's `train.py` loop looks like on
sk_gs_amd.  One process per GPU (`python -m torch.distributed.run --nproc-per-node N examples/train_views.py`) or a
single process.

    FusedViewStep   : deform network -> bone chain -> KNN / LBS weights -> skinning -> rasterize -> 0.8 L1 + 0.2 (1 - SSIM)
                      -> backward, all as direct calls into libskgs_hip.so, gradients written into the parameters' .grad
    FusedAdam       : every parameter group in one launch (eps = 1e-15, the reference's learning-rate ratios)
    ViewTable       : camera, time and target of every view as a device record; `select(v)` + ONE captured hipGraph
                      (two with > 1 rank: the all-reduce sits between them) serve all views
    OverflowGuard   : a forward whose tile lists outgrew the binning capacity is detected, the state rolled back to the
                      last snapshot, the capacity grown and the iterations redone
    ViewParallel / BucketedGradReducer / ShFactorExchange : one RCCL all-reduce (+ one small all-gather) per step
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gaussians', type=int, default=20000)
    ap.add_argument('--bones', type=int, default=12)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--iters', type=int, default=400)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--densify-every', type=int, default=0, help='clone / split / prune every N iterations (0: never)')
    ap.add_argument('--capacity', type=float, default=1.5,
                    help='one rank: row capacity as a multiple of --gaussians (sk_gs_amd/capacity.py): densification within it '
                         'happens in place and the captured step is never rebuilt or re-captured; 0 = rebuild after every event')
    args = ap.parse_args()

    from sk_gs_amd import _C, densify, scene
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.model import SkinnedGaussians
    from sk_gs_amd.optim import CapacityExceeded, FusedAdam
    from sk_gs_amd.overflow import OverflowGuard
    from sk_gs_amd.train_step import FusedTrainStep, GraphedSteps
    from sk_gs_amd.view_slot import ViewTable
    from sk_gs_amd.view_parallel import BucketedGradReducer, ShFactorExchange, ViewParallel, init_distributed

    rank, world, local_rank = init_distributed()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    P, M, W = args.gaussians, args.bones, args.size
    # "ground truth": a scene rendered from a second, perturbed set of parameters
    teacher = SkinnedGaussians(P, M, 4, num_frames=args.views, seed=1, deform_net=True).to(dev)
    model = SkinnedGaussians(P, M, 4, num_frames=args.views, seed=1, deform_net=True, learn_joints=True).to(dev)
    with torch.no_grad():
        model._features_dc.mul_(0.5)
        model._opacity.sub_(0.5)
    cams = [scene.make_camera(W, W, seed=i) for i in range(args.views)]
    views = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
    bg = torch.ones(3, device=dev)
    _C.config.sync_num_rendered = True
    with torch.no_grad():
        outs = [teacher.render(views[v], time_id=v, background=bg) for v in range(args.views)]
    table = ViewTable(views, [float(model.frame_times[v]) for v in range(args.views)], list(range(args.views)),
                      torch.stack([o['images'] for o in outs]).contiguous(), dev)
    capacity = [int(max(o['buffer'].R for o in outs) * 1.5) + 4096]  # grown by the overflow guard when it stops fitting

    if world == 1 and args.capacity and args.densify_every:  # BEFORE anything mirrors the parameters
        model.enable_capacity(int(P * args.capacity))
    opt = FusedAdam(model.param_groups(lr=args.lr), eps=1e-15)
    # update_learning_rate (train.py:140-141, gaussian_splatting.py:455-470, sk_gs.py:611-632) EVERY step, on the device: the closing
    # Adam launch of a step evaluates the schedules for the next one, captured graphs included
    opt.set_lr_schedule('xyz', lr_init=args.lr * 0.16, lr_final=args.lr * 0.0016, max_steps=30_000, lr_delay_mult=0.01)
    if any(g.get('name') == 'deform_net' for g in opt.param_groups):
        opt.set_lr_schedule('deform_net', lr_init=args.lr, lr_final=args.lr * 0.002, max_steps=40_000, lr_delay_mult=0.01)

    def build_runtime(stats_from=None):
        """everything sized by the number of Gaussians: gradient buffers, step workspaces, captured graphs.  `stats_from`: the
        step whose densification statistics the new one continues (a rebuild for capacity only; after a densification they
        start from zero)"""
        rows = model.capacity.P_cap if model.capacity is not None else model.P  # (room for the lists of a grown model)
        cap = capacity[0] * rows // P + 4096
        if world == 1:
            vp = ViewParallel(model.parameters())               # p.grad -> views of one flat buffer
            step = FusedViewStep(model, W, W, capacity=cap, background=bg, densify_stats=True, view_table=table)
            opt.rebind()                                          # the .grad tensors moved
            # forward + backward + Adam: the per-Gaussian rows' update rides on the skeleton stage's backward launch
            train = FusedTrainStep(step, opt)
            g_all = GraphedSteps(lambda _: train(), collect_garbage=False)

            def run(v):                                           # ONE graph: the view is a device record
                table.select(v)
                g_all(0)
        else:
            # what crosses the wire per step: one all-reduce of everything except the SH coefficients and the dense LBS
            # logits (those two travel as 6 + K floats per Gaussian: factors all-gathered, compact logit gradient reduced)
            vp = ViewParallel([])                                 # view schedule + densification statistics only
            wire = [p for n, p in model.named_parameters() if n not in ('_features_dc', '_features_rest', 'sp_W')]
            red = BucketedGradReducer([wire], extras=[model.P * model.K])
            for p in (model._features_dc, model._features_rest, model.sp_W):
                p.grad = torch.zeros_like(p)
            ex = ShFactorExchange(model.P, dev)
            step = FusedViewStep(model, W, W, capacity=cap, background=bg, grad_scale=1.0 / world, densify_stats=True,
                                 spw_logit_grad=red.extra_views[0], sh_factors=ex.local, view_table=table)
            opt.rebind()
            g_fb = GraphedSteps(lambda _: step.forward_backward(), collect_garbage=False)
            g_opt = GraphedSteps(lambda _: (step.sh_grads_from_factors(ex.all, 3), step.scatter_spw_grad(), opt.step()),
                                 collect_garbage=False)

            def run(v):
                table.select(v)
                g_fb(0)
                w = red.allreduce(0)
                ex.gather()
                w.wait()
                g_opt(0)     # captured at its first call, i.e. on reduced gradients (its warm-up applies real updates)
        if stats_from is not None:
            step.xyz_gradient_accum.copy_(stats_from.xyz_gradient_accum), step.denom.copy_(stats_from.denom)
            step.max_radii2D.copy_(stats_from.max_radii2D)
        else:
            step.xyz_gradient_accum.zero_(), step.denom.zero_(), step.max_radii2D.zero_()
        return vp, step, run

    vp, step, run = build_runtime()
    guard = OverflowGuard(step, opt, every=50)
    gen = torch.Generator(device=dev).manual_seed(1234)          # same samples on every rank
    it = 0
    while it < args.iters:
        run(vp.view_index(it, args.views))
        densify_now = bool(args.densify_every) and it > 0 and it % args.densify_every == 0 and it < args.iters - 1
        # (the counter is also read right before a densification: nothing may be cloned / pruned on truncated statistics)
        act = guard.check_now(it) if densify_now else guard.after_step(it)
        if act is not None:   # some forward since the last snapshot dropped splats: state is rolled back, redo from there
            capacity[0] *= 2
            vp, step, run = build_runtime(stats_from=step)  # the guard restored the statistics into the old step's tensors
            guard.rebind(step)
            if rank == 0:
                print(f'iter {it:5d}  binning capacity overflow: redoing from iteration {act[1]} with capacity x2')
            it = act[1]
            continue
        if rank == 0 and (it % 100 == 0 or it == args.iters - 1):
            l = step.loss3.tolist()                               # synchronises
            print(f'iter {it:5d}  loss {l[0]:.5f}  (L1 {l[1]:.5f}, SSIM {l[2]:.4f})  {step.status()}')
        if densify_now:
            torch.cuda.synchronize()
            t_ev = time.perf_counter()
            vp.allreduce_densify_stats(step.xyz_gradient_accum, step.denom, step.max_radii2D)
            before = model.P
            try:
                densify.densify(model, opt, step, max_grad=2e-4, extent=4.0, generator=gen)
            except CapacityExceeded:  # re-home into twice the rows, rebuild, capture again: once per capacity doubling
                model.capacity.grow(model, 2 * model.capacity.P_cap, optimizer=opt)
                vp, step, run = build_runtime(stats_from=step)
                guard.rebind(step)
                if rank == 0:
                    print(f'iter {it:5d}  row capacity grown to {model.capacity.P_cap}: runtime rebuilt, the next step re-captures')
                densify.densify(model, opt, step, max_grad=2e-4, extent=4.0, generator=gen)
            densify.prune(model, opt, step, min_opacity=0.005, extent=4.0, max_screen_size=0.25 * W)
            torch.cuda.synchronize(); t_a = time.perf_counter()
            if model.capacity is None:
                vp, step, run = build_runtime()                  # P changed: new buffers, the next step re-captures
                guard.rebind(step)
            else:
                step.reset_densify_stats()                        # in place: same buffers, same graph, it keeps replaying
            torch.cuda.synchronize(); t_b = time.perf_counter()
            guard.checkpoint(it + 1)                              # new shapes: the snapshot is taken afresh
            if rank == 0:
                how = (f'runtime rebuild {1e3 * (t_b - t_a):.1f} ms; the next step re-captures the graph' if model.capacity is None
                       else f'in place within the row capacity {model.capacity.P_cap}: no rebuild, no re-capture')
                print(f'iter {it:5d}  densify: {before} -> {model.P} Gaussians  (clone/split/prune {1e3 * (t_a - t_ev):.1f} ms; {how})')
        it += 1
    if rank == 0:
        print('visible at least once:', int((step.denom > 0).sum()), 'of', model.P, 'Gaussians; max screen radius',
              float(step.max_radii2D.max()))
    if dist.is_initialized():  # every rank applied the same reduced gradients and took the same densification decisions
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                             [torch.tensor(float(model.P), dtype=torch.float64, device=dev)])
        every = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(every, digest)
        if rank == 0:
            print('replicas identical:', all(torch.equal(every[0], e) for e in every))
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
