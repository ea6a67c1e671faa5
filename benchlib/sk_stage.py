"""bench.py --stage sk (default): the SKELETON stage of the hot path, BASELINE's headline workload.

One measurement = `run(args, env)`: the synthetic scene resident in HBM (`build_scene`), the gradient exchange of a multi-rank
run (`build_exchange`), the tile-list sizes learnt with the synchronising path (`learn_tile_lists`), the step as eager calls and
as captured hipGraphs (`build_steps`), warm-up, exactly args.steps timed steps between barriers (benchlib/timing.py), the
per-kernel pass, BASELINE's second metric (benchlib/render_protocol.py) and the JSON line.  A multi-rank run with no exchange
flag calls `run` once per exchange variant (benchlib/exchange_rank.py).
"""
import json
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist

from benchlib import render_protocol, timing
from benchlib.options import CONFIGS, alg_bytes
from benchlib.roofline import kernels_sum, render_backward_roofline


# ------------------------------------------------------------------------------------------------ scene
def build_scene(args, env):
    """the model, cameras, targets and (slot mode) the device-resident view table; everything seeded"""
    from sk_gs_amd import scene
    from sk_gs_amd.model import SkinnedGaussians
    cfg = CONFIGS[args.config]
    if cfg['M'] == 0:  # static stage (config #0): no skinning, the operator path runs it
        args.autograd = True
    P, M, K, W, H = cfg['P'], cfg['M'], cfg['K'], cfg['W'], cfg['H']
    dev, frames = env.dev, args.views
    model = SkinnedGaussians(P, M, K, sh_degree=3, num_frames=frames, seed=0, deform_net=args.deform_net,
                             scale_mult=args.scale_mult, learn_joints=args.learn_joints).to(dev)
    if not args.keep_order and M > 0:
        # Gaussians along a Z-order curve (sk_gs_amd/densify.py::sort_spatially): neighbours in space become neighbours in
        # memory and in a wavefront -- what a training loop does after a densification event (the reference's order carries no
        # meaning: clones and split children are appended, gaussian_splatting.py:577-587)
        from sk_gs_amd.densify import sort_spatially
        sort_spatially(model)
    densify_every = args.densify_every if (env.world == 1 and not args.autograd and M > 0) else 0
    if densify_every:  # room to grow BEFORE anything mirrors the parameters (gradient slots, moments, workspaces)
        model.enable_capacity(int(P * 1.25))
    cams = [scene.make_camera(W, H, seed=i) for i in range(args.views)]
    settings = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
    background = torch.ones(3, device=dev)
    # targets = the model's own initial renders + noise: a plausible fitting problem whose gradients stay small, so the workload
    # (num_rendered, tile lists) is stationary over the run instead of drifting with a random target
    gen = torch.Generator().manual_seed(77)
    targets = []
    with torch.no_grad():
        for v in range(args.views):
            if args.targets == 'uniform':  # target image U(0,1)
                targets.append(torch.rand(3, H, W, generator=gen).to(dev).contiguous())
                continue
            img = model.render(settings[v], time_id=v % frames, background=background)['images']
            targets.append((img + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1).contiguous())
    # every per-view input of the step as a device load: one captured graph serves all views (sk_gs_amd/view_slot.py)
    view_table = None
    if args.deform_net and M > 0 and not args.autograd and not args.graph_per_view:
        from sk_gs_amd.view_slot import ViewTable
        view_table = ViewTable(settings, [float(model.frame_times[v % frames]) for v in range(args.views)],
                               [v % frames for v in range(args.views)], torch.stack(targets), dev)
    return SimpleNamespace(cfg=cfg, P=P, M=M, K=K, W=W, H=H, frames=frames, model=model, settings=settings, targets=targets,
                           background=background, view_table=view_table, densify_every=densify_every)


def lr_schedules(args, opt) -> str:
    """the reference's update_learning_rate on the device (FusedAdam.set_lr_schedule): `xyz` and the deform network's group, when
    `opt` holds them.  Returns what the bench line says about it."""
    names = {g.get('name') for g in opt.param_groups}
    if args.lr_schedule != 'device' or not hasattr(opt, 'set_lr_schedule'):
        return 'off: constant rates'
    said = []
    rate = {g.get('name'): float(g['lr']) for g in opt.param_groups}
    if 'xyz' in names:  # gaussian_splatting.py:455-470: from the group's rate (lr_position_init) to 1 % of it (lr_position_final), 30k steps
        opt.set_lr_schedule('xyz', lr_init=rate['xyz'], lr_final=rate['xyz'] * 0.01, max_steps=30_000, lr_delay_mult=0.01)
        said.append(f"xyz {rate['xyz']:.3g} -> {rate['xyz'] * 0.01:.3g} over 30000 steps")
    for net in ('deform_net', 'sp_deform'):
        if net in names:  # sk_gs.py:611-614: from the group's rate to 0.0016 / (0.16 x 5) of it over lr_deform_max_steps = 40k
            opt.set_lr_schedule(net, lr_init=rate[net], lr_final=rate[net] * 0.002, max_steps=40_000, lr_delay_mult=0.01)
            said.append(f'{net} {rate[net]:.3g} -> {rate[net] * 0.002:.3g} over 40000 steps')
    return ('get_expon_lr_func evaluated per step by the closing Adam launch (update_learning_rate, train.py:140-141): '
            + '; '.join(said)) if said else 'off: no scheduled group in this optimizer'


# ------------------------------------------------------------------------------------------------ exchange (world > 1)
def build_exchange(args, env, s):
    """how the gradients cross the wire.  Default: ONE plain SUM all-reduce of the flat gradient buffer -- what the north star names
    and the easiest to trust on first contact with RCCL.  The byte-saving exchanges: --compact-logits, --sh-factors (the SH
    gradient of one view is rank-1 per Gaussian, basis(view direction) x colour gradient: the ranks exchange the two factors and
    every rank rebuilds and sums the rows in rank order), --pipeline"""
    from sk_gs_amd.view_parallel import ViewParallel
    model, P = s.model, s.P
    x = SimpleNamespace(sh_ex=None, fac_all=None, fac_local=None)
    x.fused_dist = env.use_dist and not args.autograd and not args.torch_adam
    x.pipelined = x.fused_dist and args.pipeline
    args.overlap_gather = args.overlap_gather and args.sh_factors
    x.compact = x.fused_dist and (x.pipelined or args.compact_logits or args.sh_factors)
    x.sh_factored = x.compact and not x.pipelined and args.sh_factors
    x.split_rest = bool(x.sh_factored and args.split_rest)
    x.overlap_gather = bool(x.sh_factored and args.overlap_gather)
    if not x.compact:
        x.vp = ViewParallel(model.parameters(), average=True)
        x.comm_bytes = x.vp.grads.nbytes
        return x
    # the dense [P,M] sp_W gradient never goes on the wire: the ranks all-reduce the compact [P,K] logit gradient (their KNN
    # indices are identical) and expand it afterwards
    from sk_gs_amd.view_parallel import BucketedGradReducer
    bucket0 = [model._features_dc, model._features_rest]   # final after the rasterizer backward
    bucket1 = [model._xyz, model._scaling, model._rotation, model._opacity]  # final after the skinning backward
    bucket1 += [t for t in (model.sk_r, model.sk_d_rot, model.sk_d_scale, model.global_tr) if t is not None]
    if model.sk_deform_net is not None:
        bucket1 += list(model.sk_deform_net.parameters())
    if model.learn_joints:
        bucket1.append(model.joints)
    if x.pipelined:
        x.vp = BucketedGradReducer([bucket0, bucket1], extras=[0, P * model.K])
    elif x.split_rest:
        rows_b = [model._xyz, model._scaling, model._rotation, model._opacity]
        x.vp = BucketedGradReducer([rows_b, [t for t in bucket1 if all(t is not r for r in rows_b)]], extras=[P * model.K, 0])
    elif x.sh_factored:
        x.vp = BucketedGradReducer([bucket1], extras=[P * model.K])
    else:
        x.vp = BucketedGradReducer([bucket0 + bucket1], extras=[P * model.K])
    if x.sh_factored:
        for p_ in bucket0:  # not on the wire: plain gradient tensors, rebuilt from the gathered factors
            p_.grad = torch.zeros_like(p_)
        from sk_gs_amd.view_parallel import ShFactorExchange
        x.sh_ex = ShFactorExchange(P, env.dev)
        x.fac_all, x.fac_local = x.sh_ex.all, x.sh_ex.local
    x.comm_bytes = x.vp.nbytes + (x.fac_all.numel() * 4 if x.sh_factored else 0)
    return x


# ------------------------------------------------------------------------------------------------ tile lists
def learn_tile_lists(args, s):
    """R per view, the longest tile list and the walked pairs, with the synchronising path; the bucket size of the tile lists"""
    from sk_gs_amd import _C
    _C.config.sync_num_rendered = True
    Rs, longest, walked = [], 0, []
    with torch.no_grad():  # no autograd graph may stay alive across a capture (see sk_gs_amd/train_step.py)
        for v in range(args.views):
            buf = s.model.render(s.settings[v], time_id=v % s.frames, background=s.background)['buffer']
            Rs.append(buf.R)
            longest = max(longest, _C.read_status(buf.geomBuffer)['max_tile_count'])
            # n_contrib[H, W] heads the image buffer: per pixel, how far down its tile's list the blend walked
            walked.append(int(buf.imgBuffer[:s.W * s.H * 4].view(torch.int32).sum(dtype=torch.int64)))
    s.R_mean, s.R_max, s.longest, s.walked = sum(Rs) / len(Rs), max(Rs), longest, walked
    # fixed slots per tile for the bucket layout: 1.5x the longest list seen, rounded up to 64 (overflow is counted on the device
    # and asserted to be zero after the timed region; a training loop recovers with OverflowGuard)
    hr = max(1.0, float(args.list_headroom))  # (a workload that drifts -- U(0,1) targets -- needs room to grow into)
    s.tile_bucket = 0 if args.compact_lists else ((int(longest * 1.5 * hr) + 63) // 64) * 64
    if 512 < s.tile_bucket and longest * 1.2 * hr <= 512:
        s.tile_bucket = 512  # a bucket one wave sorts needs no merge-sort launch behind it (20 % head room instead of 50)
    s.R_max = int(s.R_max * hr)
    _C.config.sync_num_rendered = False
    _C.update_capacity_hint(s.P, s.W, s.H, int(s.R_max * 1.25), 0 if args.compact_lists else int(longest * hr))  # (operator path too)


# ------------------------------------------------------------------------------------------------ the step
def build_steps(args, env, s, x):
    """the training step as eager calls (`eager_step(i)`) and as captured graphs (`graph_step(i)` after `capture_all()`), for
    every combination of one rank / view-parallel ranks, fused step / operator path, exchange variant"""
    from sk_gs_amd import _C
    from sk_gs_amd.losses import image_loss
    from sk_gs_amd.train_step import GraphedSteps
    model, vp, view_table, world = s.model, x.vp, s.view_table, env.world
    settings, targets, frames, background = s.settings, s.targets, s.frames, s.background
    compact, sh_factored, pipelined, sh_ex = x.compact, x.sh_factored, x.pipelined, x.sh_ex
    groups = model.param_groups(lr=args.lr)
    t = SimpleNamespace(fstep=None, opt=None, g_step=None, fused_update=False, train1=None, train_n=None, steps_per_graph=1,
                        train_chunk=None)
    t.overflow = torch.zeros(1, dtype=torch.int32, device=env.dev)

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

    t.fb_args, t.select, t.gkey = fb_args, select, gkey
    t.capture_views = capture_views = [vp.view_index(0, args.views)] if view_table is not None else list(range(args.views))
    if not args.autograd:
        from sk_gs_amd.fused_step import FusedViewStep
        fstep = t.fstep = FusedViewStep(
            model, s.W, s.H, capacity=int(s.R_max * 1.25 * _C.config.capacity_growth) + 1024, background=background,
            grad_scale=1.0 / world, spw_logit_grad=next(e for e in vp.extra_views if e is not None) if compact else None,
            tile_bucket=s.tile_bucket, sh_factors=x.fac_local if sh_factored else None, fused_deform_net=not args.layered_mlp,
            view_table=view_table, densify_stats=bool(s.densify_every))
        # the per-frame table gradients (one row written per step) are cleared by the Adam launch itself
        table_span = None if args.torch_adam else fstep.table_grad_span()
        fstep.tables_zeroed_by_optimizer = table_span is not None
    else:
        fstep, table_span = None, None

    if pipelined:
        # ---- world > 1: bucket 0 is on the wire while the skinning backward runs, bucket 1 while Adam updates bucket 0
        from sk_gs_amd.optim import FusedAdam
        optA = FusedAdam([g for g in groups if g['name'] in ('f_dc', 'f_rest')], eps=1e-15, betas=(0.9, 0.999))
        optB = FusedAdam([g for g in groups if g['name'] not in ('f_dc', 'f_rest')], eps=1e-15, betas=(0.9, 0.999),
                         zero_after_step=table_span)
        t.lr_schedule = lr_schedules(args, optB)

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
        t.opt = opt
        t.lr_schedule = lr_schedules(args, opt)
        if args.autograd:
            grad_params = [p for p in model.parameters() if p.requires_grad]

            def fwd_bwd(v):
                if world == 1 and not args.torch_adam:
                    # what optimizer.zero_grad() does (set_to_none): autograd then hands every gradient over as it is -- no zero
                    # fill of the flat buffer and no "+=" launch per parameter; FusedAdam gives each captured step a descriptor
                    # table with that capture's gradient addresses
                    for p in grad_params:
                        p.grad = None
                else:  # the all-reduce (and torch's captured Adam) need the gradients in place in the flat buffer
                    vp.grads.zero_()
                out = model.render(settings[v], time_id=v % frames, background=background)
                loss = image_loss(out['images'], targets[v])
                loss.backward()
                t.overflow.add_(out['buffer'].geomBuffer[4:8].view(torch.int32))
        elif args.autograd_fused:
            def fwd_bwd(v):  # the fused launches as ONE autograd node: forward half now, backward half when autograd reaches it
                (t.train1.loss if t.train1 is not None else fstep.loss)(*fb_args(v)).backward()
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

        def rebuild_exchanged_grads():
            if sh_factored:
                fstep.sh_grads_from_factors(x.fac_all, 3)
            if compact:
                fstep.scatter_spw_grad()

        def update(_=0):
            rebuild_exchanged_grads()
            opt.step()

        fused_path = not args.autograd and not args.torch_adam and not args.serial_adam and not args.autograd_fused
        if (env.use_dist and fused_path and view_table is not None and not args.select_per_step and args.pre_forward != 'off'):
            # view-parallel ranks: the update is the closing launch + the NEXT view's skeleton forward with the rows' Adam on its
            # idle CUs (instead of a full Adam launch now and a bare skeleton forward in the next step); every rank walks its
            # own views in the loop's order
            from sk_gs_amd.train_step import FusedTrainStep
            train_n = FusedTrainStep(fstep, opt, pre_forward=True, reduce_between=True)
            if train_n.pre_forward:
                t.train_n = train_n
                view_table.set_order([vp.view_index(i, args.views) for i in range(args.views)])
                train_n.prime()

                def update(_=0):  # noqa: F811
                    rebuild_exchanged_grads()
                    train_n.update()
        t.train1 = None
        if not env.use_dist and (fused_path or (args.autograd_fused and not args.torch_adam and not args.serial_adam)):
            from sk_gs_amd.train_step import FusedTrainStep
            t.train1 = train1 = FusedTrainStep(fstep, opt)
            t.fused_update = train1.fused
            if t.fused_update and view_table is not None and not args.select_per_step:
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
            if t.fused_update and not args.autograd_fused:
                t.train1(*fb_args(v))
                return
            fwd_bwd(v)
            reduce_grads()
            update()

        g_opt = None
        if x.overlap_gather:
            # graph(forward, loss, rasterizer backward) | all-gather of the factors beside graph(skinning backward) | all-reduce
            # of the rest | graph(SH rows, logit scatter, Adam)
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
        else:
            if not env.use_dist:  # whole step (fwd + bwd + Adam) is one graph per view
                if t.fused_update and args.autograd_fused:
                    # loss = train.loss(...); loss.backward(); optimizer.step() -- the step's launches behind the autograd API
                    t.g_step = GraphedSteps(lambda v: (fwd_bwd(v), opt.step()))
                elif t.fused_update:
                    t.g_step = GraphedSteps(lambda v: t.train1(*fb_args(v)))
                else:
                    t.g_step = GraphedSteps(lambda v: (fwd_bwd(v), opt.step()))
            else:                 # the RCCL all-reduce stays between two graphs
                t.g_step = GraphedSteps(fwd_bwd)
                g_opt = GraphedSteps(update)

            def graph_step(i):
                v = vp.view_index(i, args.views)
                select(v)
                t.g_step(gkey(v))
                if g_opt is not None:
                    reduce_grads()
                    g_opt(0)

            # several steps per graph replay: the one-rank fused step whose closing launch selects the next view (every piece of
            # state between two steps lives on the device; a densification only ever happens between two replays)
            n_multi = max(1, int(args.steps_per_graph))
            if (n_multi > 1 and not env.use_dist and t.fused_update and view_table is not None
                    and getattr(view_table, 'order', None) is not None and len(capture_views) == 1
                    and (not s.densify_every or s.densify_every % n_multi == 0)):
                t.steps_per_graph = n_multi

                def graph_chunk(i, n):
                    assert n == n_multi
                    t.g_step.replay(gkey(capture_views[0]), n)
                t.train_chunk = graph_chunk

            def capture_all():
                for v in capture_views:
                    select(v)
                    t.g_step.capture(gkey(v))
                if t.steps_per_graph > 1:
                    t.g_step.capture(gkey(capture_views[0]), repeat=t.steps_per_graph, warmup=0)
                if g_opt is not None:
                    # GraphedSteps.capture runs its function for real (warm-up) before recording it: the optimizer graph must see
                    # REDUCED gradients then, or every rank would apply its own view's gradient and the replicas would drift
                    # apart for good (tests/test_gpu_bench_contract.py runs two ranks and compares them)
                    reduce_grads()
                    g_opt.capture(0)

    if env.use_dist and args.graph_collectives and not args.eager:
        # ---- the collectives as nodes of the step graph (RCCL enqueues are capturable: probed with a 1-rank group on the build
        # box).  ONE graph launch per step; with the factor exchange the all-gather is a branch that runs beside the skinning
        # backward.  GraphedSteps.capture first RUNS the step for real (RCCL's lazy set-up happens there).
        assert not pipelined, '--graph-collectives: not with --pipeline'
        assert dist.get_backend() == 'nccl', '--graph-collectives needs the RCCL backend (a gloo collective synchronises the host)'

        def wait_all(*handles):
            for h in handles:
                if h is not None:
                    h.wait()

        if x.overlap_gather and x.split_rest:
            def whole_step(v):
                fstep.backward_raster(*fb_args(v))
                wg = sh_ex.gather(async_op=True)                  # | beside everything up to the update
                fstep.backward_skinning(*sk_args(v), part='rows')
                w0 = vp.allreduce(0, async_op=True)               # | rows + compact logits: beside the skeleton backward
                fstep.backward_skinning(*sk_args(v), part='skeleton')
                w1 = vp.allreduce(1, async_op=True)               # network, joints, tables: the exposed piece
                wait_all(wg, w0, w1)
                update()
        elif x.overlap_gather:
            def whole_step(v):
                fstep.backward_raster(*fb_args(v))
                wg = sh_ex.gather(async_op=True)
                fstep.backward_skinning(*sk_args(v))
                w = vp.allreduce(0, async_op=True)
                wait_all(wg, w)
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

        # several steps per replay here too when the table is ordered (every rank walks its own views by itself): the collectives
        # are nodes of the graph like the kernels
        n_multi = max(1, int(args.steps_per_graph))
        if (n_multi > 1 and view_table is not None and getattr(view_table, 'order', None) is not None and len(capture_views) == 1
                and not s.densify_every):
            t.steps_per_graph = n_multi

            def graph_chunk(i, n):  # noqa: F811
                assert n == n_multi
                g_whole.replay(gkey(capture_views[0]), n)
            t.train_chunk = graph_chunk

        def capture_all():  # noqa: F811
            for v in capture_views:
                select(v)
                g_whole.capture(gkey(v))
            if t.steps_per_graph > 1:
                g_whole.capture(gkey(capture_views[0]), repeat=t.steps_per_graph, warmup=0)

    def rewind_views():
        """ordered view table: the set-up steps consumed views; step i of the loops renders view_index(i) again, as with explicit
        selection (and the carried-over skeleton state is rebuilt for that view)"""
        if view_table is not None and getattr(view_table, 'order', None) is not None:
            view_table.rewind()
            (t.train1 if t.fused_update else t.train_n).prime()

    t.eager_step, t.graph_step, t.capture_all, t.rewind_views = eager_step, graph_step, capture_all, rewind_views
    t.train_step = eager_step if args.eager else graph_step
    return t


def describe_adam(x, t):
    closing = '; network, joints, tables + counter (+ the encoder backward of the joints) in one closing launch'
    if x.pipelined:
        return 'one launch per bucket'
    if t.train_n is not None:
        return ('after the all-reduce: closing launch (network, joints, tables, counter, next view), then the NEXT view\'s '
                'skeleton-forward launch with the per-Gaussian rows on its 224 idle CUs')
    if t.fused_update and t.train1.pre_forward:
        return ('per-Gaussian rows on the idle CUs of the two skeleton-stage launches (60 % beside the backward, 40 % beside the '
                'NEXT view\'s forward, which closes the step)' + closing)
    if t.fused_update:
        return 'per-Gaussian rows inside the skeleton stage\'s backward launch (its 224 idle CUs)' + closing
    return 'one launch after the backward'


# ------------------------------------------------------------------------------------------------ --densify-every
class DensifyEvents:
    """--densify-every: a densification event (clone + split + prune + statistics restart, in place) every N steps INSIDE the
    timed region.  Thresholds from the statistics the warm-up accumulated -- the 98th percentile of the mean screen-space gradient
    (clone / split ~2 % of the Gaussians per event) and the 2nd percentile of the opacity (prune ~2 %) -- fixed before the timed
    region."""

    def __init__(self, s, t, dev):
        from sk_gs_amd import densify as dn
        assert t.fused_update and len(t.g_step.graphs) == (2 if t.steps_per_graph > 1 else 1)  # (one step, N steps)
        self.dn, self.s, self.t = dn, s, t
        fstep, model = t.fstep, s.model
        acc, den = fstep.xyz_gradient_accum.view(-1), fstep.denom.view(-1).clamp_min(1)
        seen = fstep.denom.view(-1) > 0
        self.thr_grad = float(torch.quantile((acc / den)[seen][:1_000_000], 0.98)) if bool(seen.any()) else 1e9
        self.thr_op = float(torch.quantile(torch.sigmoid(model._opacity.detach().view(-1))[:1_000_000], 0.02))
        self.extent = 1.3 * 3 ** 0.5
        self.gen = torch.Generator(device=dev).manual_seed(1234)
        fstep.reset_densify_stats()
        self.log = dict(every=s.densify_every, events=0, P=[model.P], ms=[], max_grad=self.thr_grad, min_opacity=self.thr_op)
        self.marks = []

    def event(self, timed=True):
        """a clone / split that would outgrow the 1.25 x P row capacity (CapacityExceeded, sk_gs_amd/capacity.py) does not abort
        the run: the event is skipped, no further one is started and the line says so (`capacity_exceeded_at_event`; a training
        loop grows the storage and re-captures, examples/train_views.py)"""
        from sk_gs_amd.optim import CapacityExceeded
        if self.log.get('capacity_exceeded_at_event') is not None:
            return
        fstep, model, opt = self.t.fstep, self.s.model, self.t.opt
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # (behind the steps already queued: e1 - e0 is what the event costs on the GPU's time line -- surgery + the
        try:         # idle gaps of its two read-backs)
            self.dn.densify(model, opt, fstep, max_grad=self.thr_grad, extent=self.extent, generator=self.gen)
        except CapacityExceeded as e:
            self.log['capacity_exceeded_at_event'] = self.log['events']
            self.log['capacity_exceeded'] = str(e)[:200]
            return
        self.dn.prune(model, opt, fstep, min_opacity=self.thr_op, extent=self.extent, max_screen_size=None)
        fstep.reset_densify_stats()
        e1.record()
        if timed:
            self.log['events'] += 1
            self.log['P'].append(model.P)
            self.marks.append((e0, e1))

    def settle(self, train_step, first):
        """one untimed event: first-use initialisation (random generator, index kernels) is not a per-event cost"""
        for i in range(8):
            train_step(first + i)
        self.event(timed=False)
        self.log['P'] = [self.s.model.P]
        for i in range(8):
            train_step(first + i)
        torch.cuda.synchronize()

    def after_step(self, steps):
        every = self.s.densify_every

        def hook(i):
            if (i + 1) % every == 0 and i + 1 < steps:
                self.event()
        return hook

    def record(self):
        self.log['ms'] = [round(a.elapsed_time(b), 3) for a, b in self.marks]
        self.log['graphs_captured'] = len(self.t.g_step.graphs)
        self.log['row_capacity'] = self.s.model.capacity.P_cap
        self.log['how'] = ('clone + split + prune in place inside the timed region; the step is ONE hipGraph (and its '
                           f'{self.t.steps_per_graph}-step form) captured before the first event and never re-captured; `value` is '
                           'end-to-end')
        return self.log


# ------------------------------------------------------------------------------------------------ per-kernel table
def kernel_table(args, s, t, prof_all, passes):
    """HIP-event time, algorithmic bytes and GB/s per kernel.  The optimizer's stream (28 B per element: gradient, parameter and
    both moments read, the last three written) rides on the skeleton stage's launches; the network itself is 2.1 MB of weights
    (+ as much of gradients)"""
    from sk_gs_amd.train_step import FusedTrainStep
    model = s.model
    row_names = ('xyz', 'features_dc', 'features_rest', 'opacity', 'scaling', 'rotation', 'sp_W')
    rows_b = 28 * sum(p.numel() for n_, p in model.named_parameters() if n_.lstrip('_') in row_names)
    rest_b = 28 * sum(p.numel() for p in model.parameters()) - rows_b
    net_b = 4 * sum(p.numel() for n_, p in model.named_parameters() if 'deform_net' in n_)
    kernels = {}
    for name, (ms, n) in prof_all.items():
        b = alg_bytes(name, s.P, s.M, s.K, s.W, s.H, s.R_mean)
        if name in ('skeleton_forward', 'skeleton_backward', 'adam') and not args.autograd:
            if t.fused_update:  # one rank: the rows beside the skeleton backward (and, with pre_forward, partly beside the next forward)
                share = FusedTrainStep.ROWS_IN_BACKWARD if t.train1.pre_forward else 1.0
                b = {'skeleton_forward': net_b + (1.0 - share) * rows_b, 'skeleton_backward': 2 * net_b + share * rows_b,
                     'adam': rest_b}[name]
            elif t.train_n is not None:  # view-parallel ranks: all rows beside the next view's skeleton forward
                b = {'skeleton_forward': net_b + rows_b, 'skeleton_backward': 2 * net_b, 'adam': rest_b}[name]
            else:
                b = {'skeleton_forward': net_b, 'skeleton_backward': 2 * net_b, 'adam': rows_b + rest_b}[name]
        if name == 'preprocess_forward' and 'deform_forward' not in prof_all and getattr(t.fstep, 'deform_in_preprocess', False):
            # the skinning runs as a job of this launch (skgs_raster_inputs.deform_job): its bytes, minus the 44 B per Gaussian
            # of means / scales / rotations / opacity that are no longer re-read
            b += alg_bytes('deform_forward', s.P, s.M, s.K, s.W, s.H, s.R_mean) - 44 * s.P
        if name == 'preprocess_backward' and 'deform_backward' not in prof_all and getattr(t.fstep, 'deform_backward_in_preprocess', False):
            # likewise the skinning backward (skgs_raster_grads.deform_backward_job): minus the 44 B per Gaussian it no longer reads
            b += alg_bytes('deform_backward', s.P, s.M, s.K, s.W, s.H, s.R_mean) - 44 * s.P
        kernels[name] = timing.kernel_record(ms / n * 1e3, n / passes, b)
    if 'deform_backward' not in prof_all and getattr(t.fstep, 'deform_backward_in_preprocess', False) and 'preprocess_backward' in kernels:
        kernels['preprocess_backward']['includes'] = 'deform_backward (skinning + softmax backward, partial bone moments)'
    if 'deform_forward' not in prof_all and getattr(t.fstep, 'deform_in_preprocess', False) and 'preprocess_forward' in kernels:
        kernels['preprocess_forward']['includes'] = 'deform_forward (K nearest bones + softmax weights + skinning + activations)'
    return kernels


# ------------------------------------------------------------------------------------------------ one measurement
def run(args, env):
    """one complete measurement of the workload with the exchange / step options in `args`: builds the scene and the runtime from
    scratch (seed 0), captures, warms up, times exactly args.steps steps between barriers.  Returns the JSON line as a dict on
    rank 0, None elsewhere."""
    from sk_gs_amd import _C
    from benchlib import launch
    rank, world, dev = env.rank, env.world, env.dev
    if args.ppl:
        _C.set_pixels_per_lane(args.ppl)
    s = build_scene(args, env)
    x = build_exchange(args, env, s)
    learn_tile_lists(args, s)
    t = build_steps(args, env, s, x)
    model, view_table, fstep = s.model, s.view_table, t.fstep
    P, M, K, W, H = s.P, s.M, s.K, s.W, s.H

    t.eager_step(0)  # initialises optimizer state before any capture
    t.rewind_views()
    if not args.eager:  # every graph exists before the timed region, whatever --warmup is
        t.capture_all()
        t.rewind_views()
    if args.prime_steps > 0:  # setup: bring the fresh process to the steady state of a training run (see --prime-steps)
        i = 0
        while i < args.prime_steps:
            if t.train_chunk is not None and not args.eager and args.prime_steps - i >= t.steps_per_graph:
                t.train_chunk(i, t.steps_per_graph)
                i += t.steps_per_graph
            else:
                t.train_step(i)
                i += 1
        torch.cuda.synchronize()
        t.rewind_views()
    for i in range(args.warmup):
        t.train_step(i)
    torch.cuda.synchronize()
    if args.eager:
        _C.profile_enable(['render_backward'])
    _C.profile_collect()
    events = None
    if s.densify_every:
        events = DensifyEvents(s, t, dev)
        events.settle(t.train_step, args.warmup)
    elapsed, block_stats = timing.timed_steps(t.train_step, args.steps, args.warmup, dev,
                                              after_step=events.after_step(args.steps) if events else None,
                                              chunk=1 if args.eager else t.steps_per_graph, train_chunk=None if args.eager else t.train_chunk)
    ordered_views = view_table is not None and getattr(view_table, 'order', None) is not None
    adam_desc = describe_adam(x, t)
    if os.environ.get('SKGS_PRINT_DIGEST'):  # (diagnostics: what the timed steps trained -- tools/mode_equivalence.sh)
        torch.cuda.synchronize()
        dig = {n_: [float(p_.detach().double().sum()), float(p_.detach().double().norm())] for n_, p_ in model.named_parameters()}
        print('[digest] ' + json.dumps(dig), file=sys.stderr)
    if os.environ.get('SKGS_PRINT_LAYOUT'):  # (diagnostics: where the optimizer's arrays sit -- tools/run_spread.sh)
        try:
            opt = t.opt if hasattr(t, 'opt') else None
            rows = []
            for n_, p_ in model.named_parameters():
                st_ = (opt.state.get(p_) if opt is not None else None) or {}
                ptrs = [p_.data_ptr(), p_.grad.data_ptr() if p_.grad is not None else 0] + [v.data_ptr() for v in st_.values() if torch.is_tensor(v)]
                if p_.numel() >= 100_000:
                    rows.append(n_ + ':' + ','.join(f'{(a >> 12) & 0x1ff:03x}.{a & 0xfff:03x}' for a in ptrs))
            print('[layout] (bits 12..20).(bits 0..11) of param, grad, state...: ' + ' '.join(rows), file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            print('[layout] unavailable:', e, file=sys.stderr)
    prof = _C.profile_collect()
    _C.profile_enable([])
    replicas_identical, param_digest = timing.replicas_digest(model, world) if env.use_dist else (None, None)
    if fstep is not None:  # sticky device-side counter of forwards whose tile lists exceeded the capacity
        st = fstep.status()
        t.overflow += st['overflow_events']
        assert st.get('mlp_failed', 0) == 0, 'a fused deform-network launch gave up waiting for a workgroup: result invalid'
        mlp_one_xcd = st.get('mlp_one_xcd')
    assert int(t.overflow.item()) == 0, ('binning capacity overflow during the timed region: result invalid '
                                         + (f'(last step: {st}; tile bucket {s.tile_bucket} slots, R capacity for {s.R_max} x 1.25 x '
                                            f'{_C.config.capacity_growth})' if fstep is not None else ''))

    passes = min(args.steps, 20)
    prof_all = timing.profiled_eager_pass(_C, t.eager_step, args.warmup + args.steps, passes)
    if 'render_backward' not in prof:
        prof = prof_all
    kernels = kernel_table(args, s, t, prof_all, passes)
    if ordered_views:
        view_table.clear_order()  # the measurements below select their views explicitly
        (t.train1 if t.fused_update else t.train_n).set_pre_forward(False)

    ms_render, fps = None, None
    if args.ms_per_render is None:
        args.ms_per_render = world == 1
    if args.ms_per_render:
        from sk_gs_amd.train_step import GraphedSteps
        ms_render = render_protocol.ms_per_render(model, s.settings[0], H, W, dev)
        fused_forward = None
        if fstep is not None:  # the same forward through the fused step (direct C-ABI calls), one graph replay per render
            g_fwd = GraphedSteps(lambda v: fstep.forward(*(t.fb_args(v)[:2])))
            for v in t.capture_views:
                t.select(v)
                g_fwd.capture(t.gkey(v))

            def fused_forward(i):
                t.select(i % args.views)
                g_fwd(t.gkey(i % args.views))
        fps = render_protocol.forward_fps(model, s.settings, s.frames, s.background, fused_forward)

    cluster = launch.cluster_info(dist, torch, env.local_rank)  # (a collective: every rank)
    if rank != 0:
        return None
    cfg, pipelined, compact, sh_factored = s.cfg, x.pipelined, x.compact, x.sh_factored
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ms_step = elapsed / args.steps * 1e3
    line = {
        'metric': 'train iters/sec (deform + rasterize fwd+bwd + L1/SSIM loss + Adam; ms/render fwd+bwd beside it), '
                  f'{P // 1000}k Gaussians @{W}x{H}',
        'value': round(world * args.steps / elapsed, 3), 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'prime_steps': args.prime_steps, 'steps_per_graph': 1 if args.eager else t.steps_per_graph,
        'ms_per_step': round(ms_step, 4), 'ms_per_step_blocks': block_stats, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{cfg["name"]}: {P} Gaussians, {M} bones, K={K}, SH degree 3, {W}x{H}, '
                               f'{args.views} synthetic views, colmap=True, 1 view per rank per step'
                               + (f', DENSE variant: scales x{args.scale_mult}' if args.scale_mult != 1.0 else ''),
                   'num_rendered_mean': round(s.R_mean), 'num_rendered_max': s.R_max,
                   'tile_list_mean': round(s.R_mean / tiles, 1), 'tile_list_max': s.longest,
                   'walked_pairs_mean': round(sum(s.walked) / len(s.walked)),
                   'parallelism': f'view-parallel x{world}, ' + (
                       f'2-bucket grad all-reduce ({x.comm_bytes / 1e6:.1f} MB, SH bucket overlapped with the skinning '
                       f'backward, second bucket with Adam)' if pipelined else
                       f'flat-buffer grad all-reduce ({x.comm_bytes / 1e6:.1f} MB'
                       + (', compact LBS-logit gradient' if compact else '')
                       + (f', SH gradient as all-gathered factors ({world} x {P * 24 / 1e6:.1f} MB)' if sh_factored else '')
                       + ')'),
                   'launch': 'eager' if args.eager else (
                       f'ONE captured hipGraph for all {args.views} views (camera, time and target read from a device view slot)'
                       + (f', {t.steps_per_graph} consecutive steps per replay' if t.steps_per_graph > 1 else '')
                       if view_table is not None else f'one captured hipGraph per view ({args.views})'),
                   'view_select': ('by the closing launch of the previous step (ordered view table)' if ordered_views else
                                   'one 256-byte device-to-device copy per step') if view_table is not None else 'baked into the graphs',
                   'tile_lists': 'compact (count, scan, scatter)' if args.compact_lists
                   else f'buckets of {s.tile_bucket} slots per tile (longest list {s.longest})',
                   'joint_rotations': ('deform network (freq-encode + 8x256 MLP + heads) inside the step, '
                                       + ('one launch per layer' if args.layered_mlp else 'one persistent launch per direction'))
                   if args.deform_net and M > 0 else 'per-frame tables (test-time cache, sk_gs.py:1080-1085): NOT the '
                                                     'reference\'s training step',
                   'deform_net_on_one_xcd': None if (fstep is None or mlp_one_xcd is None) else dict(
                       mlp_one_xcd, note='[launches whose own census found the 32 network workgroups on ONE XCD (exchange kept in that L2, '
                                         'plain stores), launches so far] per direction; skgs_deform_mlp_xcd_mode / SKGS_MLP_XCD=0: blocks 0..31, '
                                         'write-through exchange (the layout before round 6)'),
                   'joints': 'trained, lr x 0.1 (sk_gs.py:607)' if model.learn_joints else 'fixed',
                   'gaussian_order': 'as generated (random)' if (args.keep_order or M == 0) else
                   'sorted along a Z-order curve (densify.sort_spatially: what a training loop does after each densification event)',
                   'adam': adam_desc,
                   'lr': args.lr, 'lr_schedule': getattr(t, 'lr_schedule', 'off'),
                   'targets': ("the model's own initial renders + N(0, 0.05^2) noise, clamped (SURVEY.md 8(d) says U(0,1): `survey_recipe` "
                               "below is that run)" if args.targets == 'own-render' else 'U(0,1) images (SURVEY.md 8(d))'),
                   'prime_steps_why': 'untimed steps after the capture: a fresh process reaches a training run\'s steady state only after '
                                      'tens of ms of work (0.375 -> 0.358 ms per step in a 20-step region)',
                   'step': 'autograd operator path' if args.autograd else (
                       ('FusedTrainStep behind the autograd API: loss = train.loss(...); loss.backward() [rows\' update on board]; optimizer.step() [closing launch]' if t.fused_update else 'FusedViewStep behind the autograd API: loss = step.loss(...); loss.backward(); optimizer.step()')
                       if args.autograd_fused else 'FusedViewStep (direct C-ABI calls)'),
                   'operator_path_backward_thread': args.backward_thread,
                   'replicas_identical': replicas_identical, 'param_digest': param_digest,
                   'cluster': cluster},
        'roofline': render_backward_roofline(prof, cfg, s.R_mean, sum(p_.numel() for p_ in model.parameters()), ms_step),
        'kernels': kernels,
        'kernels_sum': kernels_sum(kernels, cfg['name'], ms_step),
    }
    if ms_render:
        # the same pass as the sum of its kernels' HIP-event times inside the training step (no launch / sync overhead)
        ras = ('preprocess_forward', 'count_tiles', 'scan_tiles', 'scatter', 'tile_sort', 'render_forward', 'render_backward',
               'preprocess_backward')
        ms_render['kernel_sum'] = round(sum(kernels[k]['us'] * kernels[k]['launches_per_step'] for k in ras if k in kernels) / 1e3, 4)
        ms_render['backward_thread'] = args.backward_thread
        ms_render['how'] = ('operator path render() + torch.autograd.grad of (images, opacity) w.r.t. its five inputs, eager '
                            'launches, bucket tile lists, no host synchronisation; kernel_sum: the rasterizer kernels of the '
                            'fused step')
        line['ms_per_render_fwd_bwd'] = ms_render
        line['fps_forward_render'] = fps
    if events is not None:
        line['densify'] = events.record()
    if world == 1 and not args.no_cpu_baseline:
        from benchlib.cpu_baseline import cpu_baseline
        line['cpu_baseline'] = cpu_baseline(cfg, args.cpu_seconds, args.cpu_single_thread, configs=CONFIGS)
    if (world == 1 and not args.no_cpu_baseline and not args.no_survey_recipe and args.targets == 'own-render' and not args.keep_order
            and not args.autograd and not args.eager and args.scale_mult == 1.0):
        line['survey_recipe'] = survey_recipe_run(args)
        if args.config == 1 and not args.no_reference_route:
            line['reference_route'] = reference_route_runs(args)
    return line


def reference_route_runs(args) -> dict:
    """The reference's OWN iteration restated on the hooks (`bench.py --reference-loop`, benchlib/reference_loop.py: train.py:179-250 +
    framework.execute_backward) at this configuration, as numbers of their own beside the headline ("the driver line does not cover
    it"): the per-method fast paths, and SkeletonGaussianSplatting.render + the image-loss classes routed into the fused step
    (sk_gs_amd/reference_fused.py) in stage sk and stage sp.  Short child runs (this process keeps its GPU memory; nothing of it runs meanwhile).
    `sk_fused_headline_scene`: the same loop on the scene `value` is measured on (R = 0.52 M)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    out = {'what': "train iters/s of the reference's loop (outputs = model.render(t, info, background, time_id); losses = {rgb, ssim}; "
                   "sum(losses).backward(); torch.optim.Adam.step(); zero_grad(set_to_none=True)) on install_reference_hooks(): `accelerated` = "
                   "accelerate_reference(fused_render=False), `fused` = accelerate_reference(); eager; 100 untimed prime steps; synthetic scene of "
                   "benchlib/reference_loop.py (--loop-scene r5: R = 0.8 M tile instances; the headline's scene: `--loop-scene headline`)"}
    for key, extra in (('sk_accelerated', ['--reference-loop', 'accelerated', '--steps', '60']), ('sk_fused', ['--reference-loop', 'fused', '--steps', '200']),
                       ('sk_fused_headline_scene', ['--reference-loop', 'fused', '--steps', '200', '--loop-scene', 'headline']),
                       ('sp_fused', ['--stage', 'sp', '--reference-loop', 'fused', '--steps', '200'])):
        cmd = [sys.executable, bench, '--config', str(args.config), '--warmup', '10', '--lr', str(args.lr), '--views', str(args.views)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, stdin=subprocess.DEVNULL)
            if r.returncode != 0 or not r.stdout.strip():
                out[key] = {'error': f'child exited with {r.returncode}', 'stderr_tail': r.stderr.strip().splitlines()[-3:]}
                continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[key] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'prime_steps': d.get('prime_steps'),
                        'loss_last': d['config']['loss_last'], 'how': 'python bench.py ' + ' '.join(cmd[2:])}
            fr = d['config'].get('fused_route')
            if fr:
                out[key]['route'] = {'render_fused': fr['calls']['render_fused'], 'render_reference': fr['calls']['render_reference'], 'status': fr['status']}
        except Exception as e:  # noqa: BLE001  (the headline does not depend on it)
            out[key] = {'error': f'{type(e).__name__}: {e}'}
    return out


def survey_recipe_run(args) -> dict:
    """SURVEY.md 8(d)'s recipe to the letter, as a number of its own beside the headline: the same step and configuration with U(0,1)
    target images and the Gaussians left in generation order (no Z-order sort) -- a short run of bench.py in a child process (this
    process keeps its GPU memory; nothing of it runs meanwhile)"""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), '--config', str(args.config),
           '--targets', 'uniform', '--keep-order', '--steps', '20', '--warmup', '2', '--prime-steps', '0', '--no-cpu-baseline',
           '--no-ms-per-render', '--no-survey-recipe', '--list-headroom', '6', '--lr', str(args.lr), '--lr-schedule', args.lr_schedule, '--views', str(args.views)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, stdin=subprocess.DEVNULL)
        if r.returncode != 0 or not r.stdout.strip():
            return {'error': f'child exited with {r.returncode}', 'stderr_tail': r.stderr.strip().splitlines()[-3:]}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                'targets': d['config']['targets'], 'gaussian_order': d['config']['gaussian_order'], 'lr_schedule': d['config']['lr_schedule'],
                'num_rendered_mean': d['config']['num_rendered_mean'], 'tile_lists': d['config']['tile_lists'],
                'note': 'NOT a stationary workload: fitting U(0,1) noise, the deform network learns to blow the additive scale offsets up '
                        '(scales = exp(s) + d_scale, sk_gs.py:1202): num_rendered went 0.52 M -> 3.1 M and the longest tile list 430 -> 2091 '
                        'within 90 steps.  Hence the FIRST 22 steps of such a run (2 warm-up + 20 timed, no prime steps), tile lists with 6x '
                        'head room and the merge-sort launch behind the one-wave sort; the value already includes the growth',
                'how': 'python bench.py ' + ' '.join(cmd[2:])}
    except Exception as e:  # noqa: BLE001  (the headline does not depend on it)
        return {'error': f'{type(e).__name__}: {e}'}
