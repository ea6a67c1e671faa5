#!/usr/bin/env python3
"""Benchmark of the SK_GS hot path on MI355X: train iters/s (deform -> rasterize -> 0.8 L1 + 0.2 (1-SSIM) -> backward ->
[all-reduce] -> Adam(eps=1e-15)) for BASELINE.json config #1: 100k Gaussians, 20 bones, K=5, SH degree 3, 800x800.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  A "step" is one training iteration of ONE view per rank (the reference trains with
batch = 1 view, data_loader/build.py:26); with N ranks N views are processed per step (weak scaling) and `value` is
the whole-job rate N*K / t.  Inputs (Gaussians, cameras, bones, target images) are synthetic, seeded and resident in
HBM before the timed region.  Extra objects on the line:
  roofline      dominant kernel (render_backward): algorithmic bytes (40 R + 24 W H + 44 P) / HIP-event time
  kernels       per-kernel HIP-event time and algorithmic GB/s (separate short pass after the timed region)
  cpu_baseline  the CPU oracle (restatement of the reference kernels; the reference has no CPU path) on the host cores
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    0: dict(name='static-10k-400', P=10_000, M=0, K=0, W=400, H=400),
    1: dict(name='hook-like-100k-800', P=100_000, M=20, K=5, W=800, H=800),
    2: dict(name='atlas-like-200k-512', P=200_000, M=32, K=5, W=512, H=512),
    3: dict(name='mutant-like-300k-800', P=300_000, M=20, K=5, W=800, H=800),
    4: dict(name='zju-like-500k-1024', P=500_000, M=24, K=5, W=1024, H=1024),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def alg_bytes(name, P, M, K, W, H, R):
    """algorithmic bytes per launch, SURVEY.md section 8(d)"""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {
        'deform_forward': P * (88 + 4 * M),
        'deform_backward': P * (40 + 4 * K),
        'knn_bones': P * (12 + 8 * K + 4 * K),
        'preprocess_forward': 311 * P,
        'scan_tiles': 8 * T,
        'scatter': 28 * P + 12 * R,
        'tile_sort': 16 * R,
        'render_forward': 40 * R + 20 * W * H,
        'render_backward': 40 * R + 24 * W * H + 44 * P,
        'preprocess_backward': 627 * P,
        'image_loss_forward': 20 * 3 * W * H,    # reads x and y, writes the three derivative maps
        'image_loss_backward': 24 * 3 * W * H,   # reads the maps, x and y, writes dL/dx
    }.get(name)


from benchlib.cpu_baseline import cpu_baseline  # noqa: E402  (the CPU oracle timed on the host cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=1)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--ppl', type=int, default=0, help='pixels per lane of the blend kernels (0 = heuristic)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--cpu-single-thread', action='store_true',
                    help='also time the oracle on the bench workload with ONE thread (about a minute per iteration at config #1)')
    ap.add_argument('--ms-per-render', action='store_true', default=None,
                    help='also time rasterizer fwd+bwd alone (operator path, fixed upstream gradients, host-synchronised '
                         'per render: median / p10 / p90 of 50); default: on for a 1-GPU run')
    ap.add_argument('--no-ms-per-render', dest='ms_per_render', action='store_false')
    ap.add_argument('--lr', type=float, default=1e-4, help='base lr (reference 1e-3); small keeps the workload stationary')
    ap.add_argument('--torch-adam', action='store_true', help='use torch.optim.Adam(fused=True) instead of the one-launch kernel')
    ap.add_argument('--eager', action='store_true', help='issue every launch eagerly instead of replaying hipGraphs')
    ap.add_argument('--pipeline', action='store_true',
                    help='world > 1: two gradient buckets, the SH bucket on the wire during the skinning backward and the '
                         'second one during the Adam update of the first (4 graphs per step instead of 2: measured +58 us '
                         'of launch / stream-join overhead per step on one GPU, so it only pays when the all-reduce is slow)')
    ap.add_argument('--compact-logits', action='store_true',
                    help='world > 1: all-reduce the compact [P,K] LBS-logit gradient and expand it afterwards instead of '
                         'all-reducing the dense [P,M] sp_W gradient (the KNN indices are identical on every rank)')
    ap.add_argument('--sh-factors', action='store_true',
                    help='world > 1 (implies --compact-logits): all-gather the two factors of the SH gradient per view (24 B '
                         'per Gaussian and rank) and rebuild the rows on every rank instead of all-reducing the dense SH '
                         'gradient (192 B per Gaussian)')
    ap.add_argument('--exchange', choices=('auto', 'allreduce', 'factors', 'factors-overlap', 'pipeline', 'allreduce-graph',
                                           'factors-graph', 'factors-graph-split'), default='auto',
                    help='world > 1: how the gradients cross the wire.  auto (default, no other exchange flag given): time EVERY '
                         'variant in this process and report the fastest one whose replicas stayed bit-identical')
    ap.add_argument('--dense-spw-grad', action='store_true', help='(default since round 2; kept for old command lines)')
    ap.add_argument('--overlap-gather', action='store_true',
                    help='world > 1, factor exchange: split the backward graph after the rasterizer backward and run the '
                         'all-gather of the SH factors beside the skinning backward (one more graph launch per step)')
    ap.add_argument('--sh-allreduce', action='store_true', help='(default since round 2; kept for old command lines)')
    ap.add_argument('--compact-lists', action='store_true',
                    help='count -> scan -> scatter into compact tile lists (the reference layout) instead of fixed per-tile '
                         'buckets (no counting / scan launch)')
    ap.add_argument('--bone-tables', dest='deform_net', action='store_false', default=True,
                    help='read the joint rotations / d_rot / d_scale from per-frame tables (the test-time cache of '
                         'networks/sk_gs.py:1080-1085) instead of running the 8x256 deform network inside every step.  The '
                         'reference runs the network in every TRAINING step (sk_gs.py:1073-1074): that is the default here')
    ap.add_argument('--deform-net', dest='deform_net', action='store_true', help='(default) the deform network inside the step')
    ap.add_argument('--pre-forward', choices=('auto', 'on', 'off'), default='auto',
                    help='one rank, ordered views: end a step with the next view\'s skeleton-forward launch, which carries 40 %% of '
                         'the rows\' Adam update (auto: when that update is too large to hide beside the backward launch alone)')
    ap.add_argument('--serial-adam', action='store_true',
                    help='one rank: the whole Adam update as its own launch after the backward, instead of the per-Gaussian '
                         'rows\' update running inside the deform network\'s backward launch (on the 224 CUs it leaves idle)')
    ap.add_argument('--select-per-step', action='store_true',
                    help='one rank: copy the view\'s record into the slot before every replay instead of letting the closing '
                         'launch of the previous step do it')
    ap.add_argument('--fixed-joints', dest='learn_joints', action='store_false', default=True,
                    help='keep the joint positions constant; by default they are trained at 0.1 x lr as in stage sk '
                         '(networks/sk_gs.py:379,607): gradient through the kinematic chain and the network input')
    ap.add_argument('--scale-mult', type=float, default=1.0,
                    help='multiply every Gaussian\'s scale: 2.5 gives a DENSE scene (R of several million tile instances, tile '
                         'lists beyond 1024 entries: the LDS / global sort paths and long blend walks are timed); 1 = SURVEY 8d')
    ap.add_argument('--graph-per-view', action='store_true',
                    help='capture one hipGraph per view (camera, time and target baked into each) instead of ONE graph that '
                         'reads them from a device-resident view slot')
    ap.add_argument('--layered-mlp', action='store_true',
                    help='run the deform network as one launch per layer (csrc/mlp.hip) instead of the one-launch-per-direction '
                         'kernels (csrc/mlp_fused.hip)')
    ap.add_argument('--densify-every', type=int, default=0,
                    help='one rank, fused step: run a densification event (clone + split + prune, networks/gaussian_splatting.py:'
                         '565-650, thresholds calibrated so that ~2 %% of the Gaussians are cloned / split and ~2 %% pruned) every N '
                         'steps INSIDE the timed region.  The model gets a row capacity of 1.25 x P (sk_gs_amd/capacity.py): the '
                         'surgery happens in place and the ONE captured graph keeps replaying -- the reported it/s is end-to-end')
    ap.add_argument('--autograd', action='store_true',
                    help='run the step through the torch-autograd operator path (model.render + image_loss + backward) '
                         'instead of sk_gs_amd.fused_step.FusedViewStep (same kernels, no autograd glue)')
    ap.add_argument('--auto-budget', type=float, default=240.0,
                    help='world > 1, --exchange auto: seconds after which no further exchange variant is started (the ones that '
                         'finished are ranked and reported)')
    ap.add_argument('--split-rest', action='store_true',
                    help='with --sh-factors --overlap-gather --graph-collectives: the all-reduce of everything but the SH factors in '
                         'two pieces -- the per-Gaussian rows (final after the skinning backward launch) go on the wire beside the '
                         'skeleton backward, the network / joints / tables after it')
    ap.add_argument('--graph-collectives', action='store_true',
                    help='world > 1, RCCL backend: the exchange is captured INSIDE the step graph (one graph launch per step: '
                         'forward + backward, the collectives on the comm stream as a branch of the graph, update) instead of '
                         'two or three graphs with eagerly issued collectives between them')
    ap.add_argument('--backward-thread', choices=('caller', 'worker'), default='caller',
                    help="where torch autograd runs a backward (operator path only: ms/render and --autograd; the fused step "
                         "has no autograd in it).  'caller': sk_gs_amd.single_thread_backward(), what the install_as_* hooks "
                         "set; 'worker': torch's default per-device worker thread")
    ap.add_argument('--stage', choices=('sk', 'sp'), default='sk',
                    help="'sk' (default): the skeleton stage, BASELINE's headline workload.  'sp': the SUPERPOINT stage at the same "
                         "size -- 512 superpoints, 3+8-d search, sp_deform_net on 512 rows (30 k of the reference's 80 k default "
                         "steps, exps/default.yaml:12-26; benchlib/sp_stage.py)")
    ap.add_argument('--keep-order', action='store_true',
                    help='leave the synthetic Gaussians in their random order (default: sorted along a Z-order curve, '
                         'densify.sort_spatially, as after a densification event)')
    ap.add_argument('--superpoints', type=int, default=512, help='--stage sp: num_superpoints (exps/default.yaml:25)')
    ap.add_argument('--lbs-method', choices=('weighted_kernel', 'kernel', 'dist', 'W'), default='weighted_kernel',
                    help="--stage sp: the weighting of calc_LBS_weight (class default 'weighted_kernel', sk_gs.py:364; "
                         "exps/default.yaml:35 sets 'W': a dense [P,512] logit table)")
    args = ap.parse_args()
    from benchlib import launch
    if launch.needs_spawn(args.gpus):  # `python bench.py --gpus N` without a launcher: start the N ranks (or refuse), never run 1
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    import sk_gs_amd
    sk_gs_amd.single_thread_backward(args.backward_thread == 'caller')

    # stdout carries exactly ONE JSON line: everything else that any library prints to fd 1 (RCCL's version banner,
    # MIOpen notes) is diverted to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from sk_gs_amd import _C, scene
    from sk_gs_amd.losses import image_loss
    from sk_gs_amd.model import SkinnedGaussians
    from sk_gs_amd.view_parallel import ViewParallel, init_distributed

    # SKGS_FORCE_DIST=1: create a (1-rank) RCCL group so a single GPU exercises the multi-GPU code path
    if args.stage == 'sp':
        assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
        from benchlib import sp_stage
        line = sp_stage.run(args, alg_bytes, CONFIGS)
        if line is not None:
            os.write(json_fd, (json.dumps(line) + '\n').encode())
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    rank, world, local_rank = init_distributed(force=bool(os.environ.get('SKGS_FORCE_DIST')))
    use_dist = dist.is_initialized()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    _C.load_library()
    def run_workload(args):
        """one complete measurement of the workload with the exchange / step options in `args`: builds the scene and the
        runtime from scratch (seed 0), captures, warms up, times exactly args.steps steps between barriers.  Returns the
        JSON line as a dict on rank 0, None elsewhere.  (A multi-rank run with no exchange flag calls this once per
        exchange variant and reports the fastest one whose replicas stayed identical.)"""
        cfg = CONFIGS[args.config]
        if cfg['M'] == 0:  # static stage (config #0): no skinning, the operator path runs it
            args.autograd = True
        P, M, K, W, H = cfg['P'], cfg['M'], cfg['K'], cfg['W'], cfg['H']
        if args.ppl:
            _C.set_pixels_per_lane(args.ppl)

        # ---------------------------------------------------------------- synthetic scene, resident in HBM
        frames = args.views
        model = SkinnedGaussians(P, M, K, sh_degree=3, num_frames=frames, seed=0, deform_net=args.deform_net,
                                 scale_mult=args.scale_mult, learn_joints=args.learn_joints).to(dev)
        if not args.keep_order and M > 0:
            # Gaussians along a Z-order curve (sk_gs_amd/densify.py::sort_spatially): neighbours in space become neighbours in
            # memory and in a wavefront -- what a training loop does after a densification event (the reference's order carries no
            # meaning: clones and split children are appended, gaussian_splatting.py:577-587)
            from sk_gs_amd.densify import sort_spatially
            sort_spatially(model)
        densify_every = args.densify_every if (world == 1 and not args.autograd and M > 0) else 0
        if densify_every:  # room to grow BEFORE anything mirrors the parameters (gradient slots, moments, workspaces)
            model.enable_capacity(int(P * 1.25))
        cams = [scene.make_camera(W, H, seed=i) for i in range(args.views)]
        settings = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
        background = torch.ones(3, device=dev)
        # targets = the model's own initial renders + noise: a plausible fitting problem whose gradients stay small, so
        # the workload (num_rendered, tile lists) is stationary over the run instead of drifting with a random target
        gen = torch.Generator().manual_seed(77)
        targets = []
        with torch.no_grad():
            for v in range(args.views):
                img = model.render(settings[v], time_id=v % frames, background=background)['images']
                targets.append((img + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1).contiguous())
        # every per-view input of the step as a device load: one captured graph serves all views (sk_gs_amd/view_slot.py)
        view_table = None
        if args.deform_net and M > 0 and not args.autograd and not args.graph_per_view:
            from sk_gs_amd.view_slot import ViewTable
            view_table = ViewTable(settings, [float(model.frame_times[v % frames]) for v in range(args.views)],
                                   [v % frames for v in range(args.views)], torch.stack(targets), dev)
        fused_dist = use_dist and not args.autograd and not args.torch_adam
        pipelined = fused_dist and args.pipeline
        # default exchange: ONE plain SUM all-reduce of the flat gradient buffer -- what the north star names and the easiest
        # to trust on first contact with RCCL.  The byte-saving exchanges are opt-in until a multi-GPU measurement ranks them:
        # --compact-logits, --sh-factors (the SH gradient of one view is rank-1 per Gaussian, basis(view direction) x colour
        # gradient: the ranks exchange the two factors and every rank rebuilds and sums the rows in rank order), --pipeline
        args.overlap_gather = args.overlap_gather and args.sh_factors
        compact = fused_dist and (pipelined or args.compact_logits or args.sh_factors)
        sh_factored = compact and not pipelined and args.sh_factors
        groups = model.param_groups(lr=args.lr)
        split_rest = bool(sh_factored and args.split_rest)
        if compact:
            # the dense [P,M] sp_W gradient never goes on the wire: the ranks all-reduce the compact [P,K] logit gradient
            # (their KNN indices are identical) and expand it afterwards
            from sk_gs_amd.view_parallel import BucketedGradReducer
            bucket0 = [model._features_dc, model._features_rest]   # final after the rasterizer backward
            bucket1 = [model._xyz, model._scaling, model._rotation, model._opacity]  # final after the skinning backward
            bucket1 += [t for t in (model.sk_r, model.sk_d_rot, model.sk_d_scale, model.global_tr) if t is not None]
            if model.sk_deform_net is not None:
                bucket1 += list(model.sk_deform_net.parameters())
            if model.learn_joints:
                bucket1.append(model.joints)
            fac_local = fac_all = None
            if pipelined:
                vp = BucketedGradReducer([bucket0, bucket1], extras=[0, P * model.K])
            elif split_rest:
                rows_b = [model._xyz, model._scaling, model._rotation, model._opacity]
                vp = BucketedGradReducer([rows_b, [t for t in bucket1 if all(t is not r for r in rows_b)]], extras=[P * model.K, 0])
            elif sh_factored:
                vp = BucketedGradReducer([bucket1], extras=[P * model.K])
            else:
                vp = BucketedGradReducer([bucket0 + bucket1], extras=[P * model.K])
            if sh_factored:
                for p_ in bucket0:  # not on the wire: plain gradient tensors, rebuilt from the gathered factors
                    p_.grad = torch.zeros_like(p_)
                from sk_gs_amd.view_parallel import ShFactorExchange
                sh_ex = ShFactorExchange(P, dev)
                fac_all, fac_local = sh_ex.all, sh_ex.local
            comm_bytes = vp.nbytes + (fac_all.numel() * 4 if sh_factored else 0)
        else:
            vp = ViewParallel(model.parameters(), average=True)
            comm_bytes = vp.grads.nbytes
        # ---------------------------------------------------------------- learn R per view with the synchronising path
        _C.config.sync_num_rendered = True
        Rs, longest, walked = [], 0, []
        with torch.no_grad():  # no autograd graph may stay alive across a capture (see sk_gs_amd/train_step.py)
            for v in range(args.views):
                buf = model.render(settings[v], time_id=v % frames, background=background)['buffer']
                Rs.append(buf.R)
                longest = max(longest, _C.read_status(buf.geomBuffer)['max_tile_count'])
                # n_contrib[H, W] heads the image buffer: per pixel, how far down its tile's list the blend walked
                walked.append(int(buf.imgBuffer[:W * H * 4].view(torch.int32).sum(dtype=torch.int64)))
        R_mean, R_max = sum(Rs) / len(Rs), max(Rs)
        # fixed slots per tile for the bucket layout: 1.5x the longest list seen, rounded up to 64 (overflow is counted on the
        # device and asserted to be zero below; a training loop recovers with OverflowGuard)
        tile_bucket = 0 if args.compact_lists else ((int(longest * 1.5) + 63) // 64) * 64
        if 512 < tile_bucket and longest * 1.2 <= 512:
            tile_bucket = 512  # a bucket one wave sorts needs no merge-sort launch behind it (20 % head room instead of 50)
        _C.config.sync_num_rendered = False
        _C.update_capacity_hint(P, W, H, int(R_max * 1.25), 0 if args.compact_lists else longest)  # (operator path too)

        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        from sk_gs_amd.train_step import GraphedSteps

        # per-view arguments of the step calls: none when the view is read from the device-resident slot
        def fb_args(v):
            return () if view_table is not None else (settings[v], v % frames, targets[v])

        def sk_args(v):
            return () if view_table is not None else (v % frames,)

        def select(v):  # slot mode: one 256-byte device-to-device copy before the launches / the replay
            if view_table is not None and getattr(view_table, 'order', None) is None:  # (an ordered table advances by itself)
                view_table.select(v)

        def gkey(v):  # graphs are keyed by view only when the view is baked into them
            return 0 if view_table is not None else v

        capture_views = [vp.view_index(0, args.views)] if view_table is not None else list(range(args.views))
        if not args.autograd:
            from sk_gs_amd.fused_step import FusedViewStep
            fstep = FusedViewStep(model, W, H, capacity=int(R_max * 1.25 * _C.config.capacity_growth) + 1024,
                                  background=background, grad_scale=1.0 / world,
                                  spw_logit_grad=next(e for e in vp.extra_views if e is not None) if compact else None,
                                  tile_bucket=tile_bucket,
                                  sh_factors=fac_local if sh_factored else None, fused_deform_net=not args.layered_mlp,
                                  view_table=view_table, densify_stats=bool(densify_every))
            # the per-frame table gradients (one row written per step) are cleared by the Adam launch itself
            table_span = None if args.torch_adam else fstep.table_grad_span()
            fstep.tables_zeroed_by_optimizer = table_span is not None
        else:
            table_span = None

        if pipelined:
            # ---- world > 1: bucket 0 is on the wire while the skinning backward runs, bucket 1 while Adam updates bucket 0
            from sk_gs_amd.optim import FusedAdam
            optA = FusedAdam([g for g in groups if g['name'] in ('f_dc', 'f_rest')], eps=1e-15, betas=(0.9, 0.999))
            optB = FusedAdam([g for g in groups if g['name'] not in ('f_dc', 'f_rest')], eps=1e-15, betas=(0.9, 0.999),
                             zero_after_step=table_span)

            def part_a(v):
                fstep.backward_raster(*fb_args(v))

            def part_b(v):
                fstep.backward_skinning(*sk_args(v))

            def part_c2(_):
                fstep.scatter_spw_grad()
                optB.step()

            gA, gB = GraphedSteps(part_a), GraphedSteps(part_b)
            gC1, gC2 = GraphedSteps(lambda _: optA.step()), GraphedSteps(part_c2)

            def run_step(i, fa, fb, fc1, fc2, key=lambda v: v):
                v = vp.view_index(i, args.views)
                select(v)
                fa(key(v))
                w0 = vp.allreduce(0)
                fb(key(v))
                w1 = vp.allreduce(1)
                w0.wait()
                fc1(0)
                w1.wait()
                fc2(0)

            def eager_step(i):
                run_step(i, part_a, part_b, lambda _: optA.step(), part_c2)

            def graph_step(i):
                run_step(i, gA, gB, gC1, gC2, key=gkey)

            def capture_all():
                for v in capture_views:
                    select(v)
                    gA.capture(gkey(v))
                    gB.capture(gkey(v))
                # the optimizer graphs' capture warm-up applies real updates: on reduced gradients only (see below)
                for w in (vp.allreduce(0), vp.allreduce(1)):
                    if w is not None:
                        w.wait()
                gC1.capture(0)
                gC2.capture(0)
        else:
            if args.torch_adam:
                opt = torch.optim.Adam(groups, eps=1e-15, betas=(0.9, 0.999), fused=True, capturable=not args.eager)
            else:
                from sk_gs_amd.optim import FusedAdam
                opt = FusedAdam(groups, eps=1e-15, betas=(0.9, 0.999), zero_after_step=table_span)
            if args.autograd:
                grad_params = [p for p in model.parameters() if p.requires_grad]

                def fwd_bwd(v):
                    if world == 1 and not args.torch_adam:
                        # what optimizer.zero_grad() does (set_to_none): autograd then hands every gradient over as it is --
                        # no zero fill of the flat buffer and no "+=" launch per parameter; FusedAdam gives each captured
                        # step a descriptor table with that capture's gradient addresses
                        for p in grad_params:
                            p.grad = None
                    else:  # the all-reduce (and torch's captured Adam) need the gradients in place in the flat buffer
                        vp.grads.zero_()
                    out = model.render(settings[v], time_id=v % frames, background=background)
                    loss = image_loss(out['images'], targets[v])
                    loss.backward()
                    overflow.add_(out['buffer'].geomBuffer[4:8].view(torch.int32))
            else:
                def fwd_bwd(v):  # every gradient is overwritten in place: no zero fill of the flat buffer
                    fstep.forward_backward(*fb_args(v))
            prescaled = not args.autograd  # FusedViewStep seeds the backward with 1/world

            def reduce_grads():
                if compact:
                    ws = [vp.allreduce(i, async_op=sh_factored) for i in range(len(vp.bucket_views))]
                    if sh_factored:  # every rank's (direction, colour gradient) pairs; own slice already in place
                        sh_ex.gather()
                        for w in ws:
                            if w is not None:
                                w.wait()
                else:
                    vp.allreduce_grads(prescaled=prescaled)

            def update(_=0):
                if sh_factored:
                    fstep.sh_grads_from_factors(fac_all, 3)
                if compact:
                    fstep.scatter_spw_grad()
                opt.step()

            fused_update, train_n = False, None
            if (use_dist and not args.autograd and not args.torch_adam and not args.serial_adam and view_table is not None
                    and not args.select_per_step and args.pre_forward != 'off'):
                # view-parallel ranks: the update is the closing launch + the NEXT view's skeleton forward with the rows' Adam on
                # its idle CUs (instead of a full Adam launch now and a bare skeleton forward in the next step); every rank
                # walks its own views in the loop's order
                from sk_gs_amd.train_step import FusedTrainStep
                train_n = FusedTrainStep(fstep, opt, pre_forward=True, reduce_between=True)
                if train_n.pre_forward:
                    view_table.set_order([vp.view_index(i, args.views) for i in range(args.views)])
                    train_n.prime()
                    plain_update = update

                    def update(_=0):  # noqa: F811
                        if sh_factored:
                            fstep.sh_grads_from_factors(fac_all, 3)
                        if compact:
                            fstep.scatter_spw_grad()
                        train_n.update()
                else:
                    train_n = None
            if not use_dist and not args.autograd and not args.torch_adam and not args.serial_adam:
                from sk_gs_amd.train_step import FusedTrainStep
                train1 = FusedTrainStep(fstep, opt)
                fused_update = train1.fused
                if fused_update and view_table is not None and not args.select_per_step:
                    # the closing launch of a step selects the next view: the views are walked in the loop's order without a
                    # device-to-device copy in front of every replay
                    view_table.set_order([vp.view_index(i, args.views) for i in range(args.views)])
                    if args.pre_forward != 'off':
                        # ... and ends with the next view's skeleton-forward launch, which carries the rest of the rows' update
                        train1.set_pre_forward('auto' if args.pre_forward == 'auto' else True)
                        train1.prime()

            def eager_step(i):
                v = vp.view_index(i, args.views)
                select(v)
                if fused_update:
                    train1(*fb_args(v))
                    return
                fwd_bwd(v)
                reduce_grads()
                update()

            overlap_gather = sh_factored and args.overlap_gather
            if overlap_gather:
                # graph(forward, loss, rasterizer backward) | all-gather of the factors beside graph(skinning backward) |
                # all-reduce of the rest | graph(SH rows, logit scatter, Adam)
                def split_step(v, fa, fb, fc, key=lambda v: v):
                    select(v)
                    fa(key(v))
                    wg = sh_ex.gather(async_op=True)
                    fb(key(v))
                    w = vp.allreduce(0, async_op=True)
                    for h in (wg, w):
                        if h is not None:
                            h.wait()
                    fc(0)

                part_a = lambda v: fstep.backward_raster(*fb_args(v))   # noqa: E731
                part_b = lambda v: fstep.backward_skinning(*sk_args(v))  # noqa: E731
                gA, gB, gC = GraphedSteps(part_a), GraphedSteps(part_b), GraphedSteps(update)

                def eager_step(i):  # noqa: F811
                    split_step(vp.view_index(i, args.views), part_a, part_b, update)

                def graph_step(i):
                    split_step(vp.view_index(i, args.views), gA, gB, gC, key=gkey)

                def capture_all():
                    for v in capture_views:
                        select(v)
                        gA.capture(gkey(v))
                        gB.capture(gkey(v))
                    reduce_grads()  # the optimizer graph's capture warm-up applies real updates: reduced gradients only
                    gC.capture(0)
            elif not use_dist:  # whole step (fwd + bwd + Adam) is one graph per view
                if fused_update:
                    g_step = GraphedSteps(lambda v: train1(*fb_args(v)))
                else:
                    g_step = GraphedSteps(lambda v: (fwd_bwd(v), opt.step()))
                g_opt = None
            else:           # the RCCL all-reduce stays between two graphs
                g_step = GraphedSteps(fwd_bwd)
                g_opt = GraphedSteps(update)

            if not overlap_gather:
                def graph_step(i):
                    v = vp.view_index(i, args.views)
                    select(v)
                    g_step(gkey(v))
                    if g_opt is not None:
                        reduce_grads()
                        g_opt(0)

                def capture_all():
                    for v in capture_views:
                        select(v)
                        g_step.capture(gkey(v))
                    if g_opt is not None:
                        # GraphedSteps.capture runs its function for real (warm-up) before recording it: the optimizer graph
                        # must see REDUCED gradients then, or every rank would apply its own view's gradient and the replicas
                        # would drift apart for good (tests/test_gpu_bench_contract.py runs two ranks and compares them)
                        reduce_grads()
                        g_opt.capture(0)

        if use_dist and args.graph_collectives and not args.eager:
            # ---- the collectives as nodes of the step graph (RCCL enqueues are capturable: probed with a 1-rank group on the
            # build box).  ONE graph launch per step; with the factor exchange the all-gather is a branch that runs beside the
            # skinning backward.  GraphedSteps.capture first RUNS the step for real (RCCL's lazy set-up happens there).
            assert not pipelined, '--graph-collectives: not with --pipeline'
            assert dist.get_backend() == 'nccl', '--graph-collectives needs the RCCL backend (a gloo collective synchronises the host)'
            if overlap_gather and split_rest:
                def whole_step(v):
                    fstep.backward_raster(*fb_args(v))
                    wg = sh_ex.gather(async_op=True)                  # | beside everything up to the update
                    fstep.backward_skinning(*sk_args(v), part='rows')
                    w0 = vp.allreduce(0, async_op=True)               # | rows + compact logits: beside the skeleton backward
                    fstep.backward_skinning(*sk_args(v), part='skeleton')
                    w1 = vp.allreduce(1, async_op=True)               # network, joints, tables: the exposed piece
                    for h in (wg, w0, w1):
                        if h is not None:
                            h.wait()
                    update()
            elif overlap_gather:
                def whole_step(v):
                    fstep.backward_raster(*fb_args(v))
                    wg = sh_ex.gather(async_op=True)
                    fstep.backward_skinning(*sk_args(v))
                    w = vp.allreduce(0, async_op=True)
                    for h in (wg, w):
                        if h is not None:
                            h.wait()
                    update()
            else:
                def whole_step(v):
                    fwd_bwd(v)
                    reduce_grads()
                    update()
            g_whole = GraphedSteps(whole_step)

            def graph_step(i):  # noqa: F811
                v = vp.view_index(i, args.views)
                select(v)
                g_whole(gkey(v))

            def capture_all():  # noqa: F811
                for v in capture_views:
                    select(v)
                    g_whole.capture(gkey(v))

        train_step = eager_step if args.eager else graph_step

        def rewind_views():
            """ordered view table: the set-up steps below consumed views; step i of the loops renders view_index(i) again, as
            with explicit selection (and the carried-over skeleton state is rebuilt for that view)"""
            if view_table is not None and getattr(view_table, 'order', None) is not None:
                view_table.rewind()
                (train1 if fused_update else train_n).prime()

        eager_step(0)  # initialises optimizer state before any capture
        rewind_views()
        if not args.eager:  # every graph exists before the timed region, whatever --warmup is
            capture_all()
            rewind_views()
        for i in range(args.warmup):
            train_step(i)
        torch.cuda.synchronize()
        if args.eager:
            _C.profile_enable(['render_backward'])
        _C.profile_collect()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        # --densify-every: thresholds from the statistics the warm-up accumulated -- the 98th percentile of the mean
        # screen-space gradient (clone / split ~2 % of the Gaussians per event) and the 2nd percentile of the opacity (prune
        # ~2 %) -- fixed before the timed region; an event = densify + prune + statistics restart, in place
        densify_log = None
        if densify_every:
            from sk_gs_amd import densify as dn
            assert fused_update and len(g_step.graphs) == 1
            acc, den = fstep.xyz_gradient_accum.view(-1), fstep.denom.view(-1).clamp_min(1)
            seen = fstep.denom.view(-1) > 0
            thr_grad = float(torch.quantile((acc / den)[seen][:1_000_000], 0.98)) if bool(seen.any()) else 1e9
            thr_op = float(torch.quantile(torch.sigmoid(model._opacity.detach().view(-1))[:1_000_000], 0.02))
            extent = 1.3 * 3 ** 0.5
            dgen = torch.Generator(device=dev).manual_seed(1234)
            fstep.reset_densify_stats()
            densify_log = dict(every=densify_every, events=0, P=[model.P], ms=[], max_grad=thr_grad, min_opacity=thr_op)

            ev_marks = []

            def densify_event(timed=True):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()  # (behind the steps already queued: e1 - e0 is what the event costs on the GPU's time line --
                dn.densify(model, opt, fstep, max_grad=thr_grad, extent=extent, generator=dgen)  # surgery + the idle gaps of
                dn.prune(model, opt, fstep, min_opacity=thr_op, extent=extent, max_screen_size=None)  # its two read-backs)
                fstep.reset_densify_stats()
                e1.record()
                if timed:
                    densify_log['events'] += 1
                    densify_log['P'].append(model.P)
                    ev_marks.append((e0, e1))

            # one untimed event: first-use initialisation (random generator, index kernels) is not a per-event cost
            for i in range(8):
                train_step(args.warmup + i)
            densify_event(timed=False)
            densify_log['P'] = [model.P]
            for i in range(8):
                train_step(args.warmup + i)
            torch.cuda.synchronize()
        # the timed region: exactly args.steps steps between barriers; an event every steps/10 steps splits it into >= 10 blocks
        # (when steps >= 10) whose per-step times give the spread of `ms_per_step` (median / p10 / p90)
        n_blocks = min(args.steps, 10)
        edges = [round(b * args.steps / n_blocks) for b in range(n_blocks + 1)]
        marks = [torch.cuda.Event(enable_timing=True) for _ in edges]
        t0 = time.perf_counter()
        marks[0].record()
        nxt = 1
        for i in range(args.steps):
            train_step(args.warmup + i)
            if densify_every and (i + 1) % densify_every == 0 and i + 1 < args.steps:
                densify_event()
            if i + 1 == edges[nxt]:
                marks[nxt].record()
                nxt += 1
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        per_step = sorted(marks[b].elapsed_time(marks[b + 1]) / (edges[b + 1] - edges[b]) for b in range(n_blocks))
        block_stats = dict(blocks=n_blocks, median=round(per_step[n_blocks // 2], 4), p10=round(per_step[n_blocks // 10], 4),
                           p90=round(per_step[min(n_blocks - 1, (9 * n_blocks) // 10)], 4),
                           min=round(per_step[0], 4), max=round(per_step[-1], 4),
                           how='HIP events on the launch stream every steps/blocks steps inside the timed region (rank 0)')
        ordered_views = view_table is not None and getattr(view_table, 'order', None) is not None
        closing = '; network, joints, tables + counter (+ the encoder backward of the joints) in one closing launch'
        if pipelined:
            adam_desc = 'one launch per bucket'
        elif train_n is not None:
            adam_desc = ('after the all-reduce: closing launch (network, joints, tables, counter, next view), then the NEXT view\'s '
                         'skeleton-forward launch with the per-Gaussian rows on its 224 idle CUs')
        elif fused_update and train1.pre_forward:
            adam_desc = ('per-Gaussian rows on the idle CUs of the two skeleton-stage launches (60 % beside the backward, 40 % beside '
                         'the NEXT view\'s forward, which closes the step)' + closing)
        elif fused_update:
            adam_desc = 'per-Gaussian rows inside the skeleton stage\'s backward launch (its 224 idle CUs)' + closing
        else:
            adam_desc = 'one launch after the backward'
        prof = _C.profile_collect()
        _C.profile_enable([])
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # view-parallel replicas must stay bit-identical: same reduced gradients, same Adam step on every rank
        replicas_identical, param_digest = None, None
        if use_dist:
            names = [n for n, _ in model.named_parameters()]
            digest = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                                 [p.detach().double().abs().sum() for p in model.parameters()])
            every = [torch.empty_like(digest) for _ in range(world)]
            dist.all_gather(every, digest)
            differ = sorted({names[i % len(names)] for e in every for i in (every[0] != e).nonzero().flatten().tolist()})
            replicas_identical = True if not differ else differ
            param_digest = float(digest[len(names):].sum())  # sum |p| over all parameters (rank 0): compares exchange modes
        if not args.autograd:  # sticky device-side counter of forwards whose tile lists exceeded the capacity
            st = fstep.status()
            overflow += st['overflow_events']
            assert st.get('mlp_failed', 0) == 0, 'a fused deform-network launch gave up waiting for a workgroup: result invalid'
        assert int(overflow.item()) == 0, 'binning capacity overflow during the timed region: result invalid'

        # ------------------------------------------------ per-kernel HIP-event timing: eager pass over the same steps
        # (events cannot be read back from inside a replayed graph; the kernels and their inputs are the same)
        kernels = {}
        _C.profile_enable(None)
        for i in range(min(args.steps, 20)):
            eager_step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        prof_all = _C.profile_collect()
        if 'render_backward' not in prof:
            prof = prof_all
        for name, (ms, n) in prof_all.items():
            us = ms / n * 1e3
            b = alg_bytes(name, P, M, K, W, H, R_mean)
            if name in ('skeleton_forward', 'skeleton_backward', 'adam') and not args.autograd:
                # the optimizer's stream (28 B per element: gradient, parameter and both moments read, the last three
                # written) rides on these launches; the network itself is 2.1 MB of weights (+ as much of gradients)
                rows_b = 28 * sum(p.numel() for n_, p in model.named_parameters() if n_.lstrip('_') in
                                  ('xyz', 'features_dc', 'features_rest', 'opacity', 'scaling', 'rotation', 'sp_W'))
                rest_b = 28 * sum(p.numel() for p in model.parameters()) - rows_b
                net_b = 4 * sum(p.numel() for n_, p in model.named_parameters() if 'deform_net' in n_)
                from sk_gs_amd.train_step import FusedTrainStep as _FTS
                fu, tn = bool(locals().get('fused_update')), locals().get('train_n')
                if fu:    # one rank: the rows beside the skeleton backward (and, with pre_forward, partly beside the next forward)
                    share = _FTS.ROWS_IN_BACKWARD if train1.pre_forward else 1.0
                    b = {'skeleton_forward': net_b + (1.0 - share) * rows_b, 'skeleton_backward': 2 * net_b + share * rows_b,
                         'adam': rest_b}[name]
                elif tn is not None:  # view-parallel ranks: all rows beside the next view's skeleton forward
                    b = {'skeleton_forward': net_b + rows_b, 'skeleton_backward': 2 * net_b, 'adam': rest_b}[name]
                else:
                    b = {'skeleton_forward': net_b, 'skeleton_backward': 2 * net_b, 'adam': rows_b + rest_b}[name]
            kernels[name] = dict(us=round(us, 2), launches_per_step=round(n / min(args.steps, 20), 2),
                                 alg_MB=round(b / 1e6, 2) if b else None,
                                 GBps=round(b / (us * 1e-6) / 1e9, 1) if b else None,
                                 frac=round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if b else None)
        _C.profile_enable([])
        if ordered_views:
            view_table.clear_order()  # the measurements below select their views explicitly
            (train1 if fused_update else train_n).set_pre_forward(False)

        # ---- BASELINE's second metric and the reference's FPS protocol (test.py:56-81,102-123: warm-up, then N renders between
        # two events): 20 warm-up + 200 timed iterations, HIP events on the launch stream
        ms_render, fps = None, None
        if args.ms_per_render is None:
            args.ms_per_render = world == 1
        if args.ms_per_render:
            from sk_gs_amd.renderer.gaussian_render import render
            NW, NT = 20, 200
            with torch.no_grad():
                net = {k: v.detach() for k, v in model(0).items()}
            gcol, gop = torch.randn(3, H, W, device=dev), torch.randn(H, W, device=dev)
            ins = {k: v.clone().requires_grad_(True) for k, v in net.items()}
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NT)]
            leaves = list(ins.values())
            for i in range(NW + NT):  # rasterizer forward + backward with fixed upstream gradients, operator path, no host sync
                if i >= NW:
                    ev[i - NW][0].record()
                o = render(**ins, raster_settings=settings[0])
                # (autograd.grad: the operator's gradients are RETURNED, not accumulated into leaf .grad tensors -- five
                # AccumulateGrad add kernels per iteration, 19 MB of them the SH gradient, are not the rasterizer)
                torch.autograd.grad([o['images'], o['opacity']], leaves, [gcol, gop], allow_unused=True)
                if i >= NW:
                    ev[i - NW][1].record()
            torch.cuda.synchronize()
            times = sorted(a.elapsed_time(b_) for a, b_ in ev)
            ms_render = dict(median=round(times[NT // 2], 4), p10=round(times[NT // 10], 4), p90=round(times[9 * NT // 10], 4),
                             protocol=f'{NW} warm-up + {NT} timed iterations, one HIP event pair per iteration')
            # the same operator-path calls captured once and replayed (config.sync_num_rendered is off: nothing in them touches
            # the host): what the drop-in boundary costs without the Python / launch latency of the eager loop above
            try:
                def op_fwd_bwd(_):
                    o_ = render(**ins, raster_settings=settings[0])
                    torch.autograd.grad([o_['images'], o_['opacity']], leaves, [gcol, gop], allow_unused=True)
                del o
                g_op = GraphedSteps(op_fwd_bwd)
                g_op.capture(0)
                for i in range(NW + NT):
                    if i >= NW:
                        ev[i - NW][0].record()
                    g_op(0)
                    if i >= NW:
                        ev[i - NW][1].record()
                torch.cuda.synchronize()
                times = sorted(a.elapsed_time(b_) for a, b_ in ev)
                ms_render['graph_replay_median'] = round(times[NT // 2], 4)
                del g_op
            except Exception as e:  # noqa  (a capture problem must not cost the headline line)
                ms_render['graph_replay_median'] = None
                ms_render['graph_replay_error'] = str(e)[:200]
            # forward-only render rate, the reference's FPS (deform network + skinning + rasterize + background, no_grad)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.no_grad():
                for i in range(NW + NT):
                    if i == NW:
                        e0.record()
                    model.render(settings[i % args.views], time_id=i % frames, background=background)
                e1.record()
            torch.cuda.synchronize()
            fps = dict(operator_path=round(NT * 1000.0 / e0.elapsed_time(e1), 1),
                       protocol=f'test.py:102-123: {NW} warm-up + {NT} renders between two events, views cycled')
            if not args.autograd:  # the same forward through the fused step (direct C-ABI calls), one graph replay per render
                g_fwd = GraphedSteps(lambda v: fstep.forward(*(fb_args(v)[:2])))
                for v in capture_views:
                    select(v)
                    g_fwd.capture(gkey(v))
                for i in range(NW + NT):
                    if i == NW:
                        e0.record()
                    select(i % args.views)
                    g_fwd(gkey(i % args.views))
                e1.record()
                torch.cuda.synchronize()
                fps['fused_step_graph'] = round(NT * 1000.0 / e0.elapsed_time(e1), 1)

        from benchlib import launch as _launch
        cluster = _launch.cluster_info(dist, torch, local_rank)  # (a collective: every rank)
        if rank == 0:
            rb_ms, rb_n = prof.get('render_backward', (0.0, 0))
            rb_us = rb_ms / max(rb_n, 1) * 1e3
            rb_bytes = alg_bytes('render_backward', P, M, K, W, H, R_mean)
            achieved = rb_bytes / (rb_us * 1e-6) / 1e9 if rb_us > 0 else 0.0
            # counters are not collected in this run: what the last committed PMC profile of this kernel says goes under
            # `from_profile`, with the file and the commit it was taken at (tools/pmc_summary.py writes both)
            from_profile = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_render_backward.json')
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    if rec.get('config') == cfg['name']:
                        from_profile = {k: rec.get(k) for k in ('file', 'commit', 'hbm_bytes_per_launch', 'valubusy',
                                                                'valuutilization', 'valu_insts_per_launch', 'avg_us') if k in rec}
                        if rec.get('valu_insts_per_launch') and rec.get('avg_us'):
                            # VALU issue roofline: wave-instructions per second against 256 CUs x 4 SIMDs x one wave64
                            # instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32)
                            peak = 256 * 4 * 2.4e9 / 2
                            ach = rec['valu_insts_per_launch'] / (rec['avg_us'] * 1e-6)
                            from_profile['valu'] = dict(
                                bound='valu', achieved=round(ach / 1e9, 1), peak=round(peak / 1e9, 1),
                                unit='G wave-instructions/s', frac=round(ach / peak, 4),
                                # measured on this chip (tools/micro/valu_issue_rate.hip, profiles/*_valu_issue_rate.txt): plain
                                # fp32 / integer ops ~1000 G/s, DPP / compares / selects ~570, permlane swaps, exp, rcp ~300
                                peak_measured_plain=1000.0, frac_of_measured=round(ach / 1000.0e9, 4),
                                busy=round(rec['valubusy'] / 100.0, 4) if rec.get('valubusy') else None,
                                note='busy = share of the kernel\'s time its SIMDs spend issuing VALU work: the distance from the '
                                     'ceiling of ITS OWN instruction mix (40 % of the issue clocks of a visit are the cross-lane '
                                     'reduction: 17 DPP adds, 2 permlane swaps)')
                except Exception:
                    from_profile = None
            # the whole step against the HBM roofline: SURVEY 8(d)'s B_alg (one render's algorithmic bytes) + the optimizer's
            # 28 B per parameter element, over the measured step
            T_ = ((W + 15) // 16) * ((H + 15) // 16)
            b_alg = P * (1138 + 4 * M + 4 * K) + 116 * R_mean + 44 * W * H + 8 * T_
            b_adam = 28 * sum(p_.numel() for p_ in model.parameters())
            ms_step_ = elapsed / args.steps * 1e3
            whole_step = dict(alg_bytes_render=int(b_alg), alg_bytes_adam=int(b_adam), ms=round(ms_step_, 4),
                              frac=round((b_alg + b_adam) / (ms_step_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                              frac_render_only=round(b_alg / (ms_step_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                              note='B_alg of SURVEY 8(d) (+ 28 B per optimizer element) / ms_per_step / 8 TB/s')
            line = {
                'metric': 'train iters/sec (deform + rasterize fwd+bwd + L1/SSIM loss + Adam; ms/render fwd+bwd beside it), '
                          f'{P // 1000}k Gaussians @{W}x{H}',
                'value': round(world * args.steps / elapsed, 3), 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'ms_per_step_blocks': block_stats,
                'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': f'{cfg["name"]}: {P} Gaussians, {M} bones, K={K}, SH degree 3, {W}x{H}, '
                                       f'{args.views} synthetic views, colmap=True, 1 view per rank per step'
                                       + (f', DENSE variant: scales x{args.scale_mult}' if args.scale_mult != 1.0 else ''),
                           'num_rendered_mean': round(R_mean), 'num_rendered_max': R_max,
                           'tile_list_mean': round(R_mean / (((W + 15) // 16) * ((H + 15) // 16)), 1), 'tile_list_max': longest,
                           'walked_pairs_mean': round(sum(walked) / len(walked)),
                           'parallelism': f'view-parallel x{world}, ' + (
                               f'2-bucket grad all-reduce ({comm_bytes / 1e6:.1f} MB, SH bucket overlapped with the skinning '
                               f'backward, second bucket with Adam)' if pipelined else
                               f'flat-buffer grad all-reduce ({comm_bytes / 1e6:.1f} MB'
                               + (', compact LBS-logit gradient' if compact else '')
                               + (f', SH gradient as all-gathered factors ({world} x {P * 24 / 1e6:.1f} MB)' if sh_factored else '')
                               + ')'),
                           'launch': 'eager' if args.eager else (
                               f'ONE captured hipGraph for all {args.views} views (camera, time and target read from a device view slot)'
                               if view_table is not None else f'one captured hipGraph per view ({args.views})'),
                           'view_select': ('by the closing launch of the previous step (ordered view table)' if ordered_views else
                                           'one 256-byte device-to-device copy per step') if view_table is not None else 'baked into the graphs',
                           'tile_lists': 'compact (count, scan, scatter)' if args.compact_lists
                           else f'buckets of {tile_bucket} slots per tile (longest list {longest})',
                           'joint_rotations': ('deform network (freq-encode + 8x256 MLP + heads) inside the step, '
                                               + ('one launch per layer' if args.layered_mlp else 'one persistent launch per direction'))
                           if args.deform_net and M > 0 else 'per-frame tables (test-time cache, sk_gs.py:1080-1085): NOT the '
                                                             'reference\'s training step',
                           'joints': 'trained, lr x 0.1 (sk_gs.py:607)' if model.learn_joints else 'fixed',
                           'adam': adam_desc,
                           'step': 'autograd operator path' if args.autograd else 'FusedViewStep (direct C-ABI calls)',
                           'operator_path_backward_thread': args.backward_thread,
                           'replicas_identical': replicas_identical, 'param_digest': param_digest,
                           'cluster': cluster},
                'roofline': {'bound': 'hbm', 'kernel': 'render_backward', 'achieved': round(achieved, 2),
                             'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 5),
                             # HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, separate passes): not
                             # collected in this run -- the figure of the last committed counter profile of this kernel on this
                             # workload, with the file and commit it comes from (null when there is none for this workload)
                             'traffic': (from_profile or {}).get('hbm_bytes_per_launch'),
                             'traffic_source': ({k_: from_profile.get(k_) for k_ in ('file', 'commit')} if from_profile else None),
                             'whole_step': whole_step,
                             'avg_us': round(rb_us, 2), 'launches': rb_n,
                             'alg_bytes_per_launch': int(rb_bytes), 'limiter': 'valu',
                             'from_profile': from_profile,
                             'note': 'the dominant kernel is VALU-issue bound, not HBM-bound (SURVEY 8d caveat): the HBM '
                                     'fraction above is the contract figure, from_profile.valu the one that bounds it; '
                                     'streaming kernels are listed under "kernels" with their GB/s; traffic is null because no '
                                     'counter is collected in this run (see from_profile)'},
                'kernels': kernels,
            }
            if ms_render:
                # the same pass as the sum of its kernels' HIP-event times inside the training step (no launch / sync overhead)
                ras = ('preprocess_forward', 'count_tiles', 'scan_tiles', 'scatter', 'tile_sort', 'render_forward',
                       'render_backward', 'preprocess_backward')
                ms_render['kernel_sum'] = round(sum(kernels[k]['us'] * kernels[k]['launches_per_step']
                                                    for k in ras if k in kernels) / 1e3, 4)
                ms_render['backward_thread'] = args.backward_thread
                ms_render['how'] = 'operator path render() + torch.autograd.grad of (images, opacity) w.r.t. its five inputs, eager ' \
                                   'launches, bucket tile lists, no host synchronisation; kernel_sum: the rasterizer kernels of ' \
                                   'the fused step'
                line['ms_per_render_fwd_bwd'] = ms_render
                line['fps_forward_render'] = fps
            if densify_log is not None:
                densify_log['ms'] = [round(a_.elapsed_time(b_), 3) for a_, b_ in ev_marks]
                densify_log['graphs_captured'] = len(g_step.graphs)
                densify_log['row_capacity'] = model.capacity.P_cap
                densify_log['how'] = ('clone + split + prune in place inside the timed region; the step is ONE hipGraph captured '
                                      'before the first event and never re-captured; `value` is end-to-end')
                line['densify'] = densify_log
            if world == 1 and not args.no_cpu_baseline:
                line['cpu_baseline'] = cpu_baseline(cfg, args.cpu_seconds, args.cpu_single_thread, configs=CONFIGS)
            return line
        return None

    # ---------------------------------------------------------------- which exchange?  (world > 1)
    # A multi-rank run with no exchange flag times every exchange variant -- same scene, same seed, the runtime rebuilt for
    # each, exactly args.steps steps between barriers each -- and reports the FASTEST ONE WHOSE REPLICAS STAYED IDENTICAL as
    # `value`; all of them are listed under `exchange_variants`.  No multi-GPU box exists in the build loop: the ranking is
    # done where the xGMI links are (DESIGN.md section 6 holds the predicted table to read the record against).
    import copy
    import gc
    explicit = (args.pipeline or args.compact_logits or args.sh_factors or args.overlap_gather or args.graph_collectives or args.split_rest
                or args.exchange != 'auto')
    # order = the order they are timed in: the predicted-best variant with EAGER collectives first (`factors`
    # 5.9x, `factors-overlap` 6.0x against 4.7x for the plain all-reduce), so a caller's time limit that cuts the ranking
    # short still records it; the plain all-reduce second; the captured-collective variants last, each under the watchdog
    variants = {
        'factors': dict(sh_factors=True, compact_logits=True),
        'allreduce': dict(),
        'factors-overlap': dict(sh_factors=True, compact_logits=True, overlap_gather=True),
        'pipeline': dict(pipeline=True),
        # the same two exchanges with the collectives captured inside ONE step graph (last: a fabric on which a captured
        # collective never completes costs only these two entries, see bail())
        'allreduce-graph': dict(graph_collectives=True),
        'factors-graph': dict(sh_factors=True, compact_logits=True, overlap_gather=True, graph_collectives=True),
        'factors-graph-split': dict(sh_factors=True, compact_logits=True, overlap_gather=True, graph_collectives=True,
                                    split_rest=True),
    }
    if args.exchange != 'auto':
        for k_, v_ in variants[args.exchange].items():
            setattr(args, k_, v_)
    if world > 1 and not explicit and not args.autograd and not args.torch_adam:
        import threading
        lines, errors = {}, {}

        def emit(aborted=None):
            """rank 0: the fastest variant whose replicas stayed identical, all of them under `exchange_variants`"""
            if rank != 0:
                return
            summary = {}
            for name in variants:
                ln = lines.get(name)
                if ln is None:
                    summary[name] = dict(error=errors.get(name, 'abandoned: it did not finish in time' if name == aborted
                                                          else 'not run'))
                else:
                    summary[name] = dict(value=ln['value'], ms_per_step=ln['ms_per_step'],
                                         ms_per_step_blocks=ln.get('ms_per_step_blocks'),
                                         parallelism=ln['config']['parallelism'],
                                         replicas_identical=ln['config']['replicas_identical'],
                                         param_digest=ln['config']['param_digest'])
            good = [n for n in variants if lines.get(n) is not None and lines[n]['config']['replicas_identical'] is True]
            assert good, f'no exchange variant kept the replicas identical: {summary}'
            best = max(good, key=lambda n: lines[n]['value'])
            line = lines[best]
            line['config']['exchange'] = best
            line['exchange_variants'] = summary
            os.write(json_fd, (json.dumps(line) + '\n').encode())

        def bail(name):
            # a later variant hangs (a collective that never completes on this fabric) or dies: the record must not die with
            # it.  Every rank's own timer ends its process; rank 0 first prints what the finished variants measured.
            try:
                emit(aborted=name)
            finally:
                os._exit(0)

        t_first = None
        t_auto0 = time.perf_counter()
        for name, flags in variants.items():
            # a wall-clock budget for the whole ranking: the record is ONE line at the very end, so a caller's time limit that
            # fell in the middle of a late variant would cost all of it.  Every rank takes the same decision (MAX of the clocks).
            if any(n_ not in errors for n_ in lines):  # (the same on every rank: errors are agreed on below; lines are rank 0's)
                el = torch.tensor([time.perf_counter() - t_auto0], dtype=torch.float64, device=dev)
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
                if float(el.item()) > args.auto_budget:
                    errors[name], lines[name] = f'not run: the ranking had used {float(el.item()):.0f} s of its {args.auto_budget:.0f} s budget', None
                    continue
            a = copy.copy(args)
            for k_, v_ in flags.items():
                setattr(a, k_, v_)
            a.ms_per_render, a.no_cpu_baseline = False, True
            have_one = any(v is not None for v in lines.values())
            timer = None
            if have_one:  # (the first variant -- the plain all-reduce -- runs unguarded: without it there is no record)
                timer = threading.Timer(max(180.0, 6.0 * (t_first or 30.0)), bail, args=(name,))
                timer.daemon = True
                timer.start()
            t_v = time.perf_counter()
            try:
                try:
                    lines[name] = run_workload(a)
                except Exception as e:  # noqa  (a variant that cannot run here must not cost the record)
                    errors[name] = f'{type(e).__name__}: {e}'[:300]
                    lines[name] = None
                # every rank must agree on whether the variant ran (an exception on one rank only would desynchronise the next)
                ok = torch.tensor([0 if name in errors else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and name not in errors:
                    errors[name], lines[name] = 'failed on another rank', None
                gc.collect()
                torch.cuda.empty_cache()
                dist.barrier()
            except Exception as e:  # noqa  (the process group itself is broken: report what is there and stop)
                errors.setdefault(name, f'{type(e).__name__}: {e}'[:300])
                if timer is not None:
                    timer.cancel()
                if any(v is not None for v in lines.values()):
                    bail(name)
                raise
            finally:
                if timer is not None:
                    timer.cancel()
            if t_first is None:
                t_first = time.perf_counter() - t_v
        emit()
    else:
        line = run_workload(args)
        if rank == 0:
            line['config']['exchange'] = (('pipeline' if args.pipeline else 'factors-overlap' if args.overlap_gather else
                                           'factors' if args.sh_factors else 'compact-logits' if args.compact_logits else
                                           'allreduce') + ('-graph' if args.graph_collectives else '') + ('-split' if args.split_rest else '')) if world > 1 else None
            os.write(json_fd, (json.dumps(line) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()

if __name__ == '__main__':
    main()
