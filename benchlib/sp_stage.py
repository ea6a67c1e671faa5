"""bench.py --stage sp: the SUPERPOINT stage of the hot path at BASELINE config #1's size.

Stage `sp` is 30 k of the reference's 80 k default steps (exps/default.yaml:12-19: sp_fix 3 000 + sp 27 000): 100k Gaussians
skinned by M = 512 superpoints (num_superpoints, :25) found by a K = 5 nearest search over [xyz | 8 hyper coordinates]
(hyper_dim 8, :26; sk_gs.py:751-757), `sp_deform_net` (DeformNetwork, 8 x 256, on the 512 superpoints; sk_gs.py:209-315) producing
their transforms every step, then the same rasterizer, loss and Adam as the skeleton stage.  One step =

    sp net forward (1 launch, MFMA) | search + weighting (1) | skinning (1) | preprocess | scatter | sort | blend forward |
    loss forward | loss backward | blend backward | preprocess backward | skinning backward | weighting backward (2) |
    sp net backward (2; the per-Gaussian rows' Adam on the first one's idle CUs) | closing optimizer launch

as ONE captured hipGraph for all views (camera, time, target from the device view slot).  Returns the JSON line as a dict.
"""
import torch
import torch.distributed as dist

from benchlib import timing
from benchlib.options import HBM_PEAK_GBPS

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def net_flops(M: int):
    fwd = 2 * M * (93 * 256 + 6 * 256 * 256 + 349 * 256 + 256 * 10) + 2 * (13 * 256 + 256 * 30)
    return fwd, 2 * fwd


def alg_bytes_sp(name, P, M, K, F, W, H, R, rows_adam_bytes=0):
    """algorithmic bytes per launch of the stage's own kernels (the rasterizer's are bench.alg_bytes)"""
    net_w = 4 * (13 * 256 + 256 + 256 * 30 + 30 + 93 * 256 + 6 * 256 * 256 + 349 * 256 + 8 * 256 + 10 * 256 + 10)
    acts = 4 * M * (96 + 8 * 256)
    return {
        # positions + hyper row in, K (index 8 B, weight 4 B, distance 4 B) out; the superpoint table is LDS-resident
        'sp_knn_weights': P * (12 + 4 * F + 16 * K) + M * (12 + 4 * F + 8),
        # K (weight, index, distance, cotangent) in, hyper gradient row out + per-workgroup partial tables
        'sp_knn_weights_backward': P * (20 * K + 8 * F),
        'sp_net_forward': net_w + acts + M * (12 + 4 * (10 + 7 + 4 + 3)),       # weights once, saved activations out
        'sp_net_backward': 3 * net_w + 3 * acts + rows_adam_bytes,               # weights, activations, gZ out + in, gradients out
        'deform_forward': P * (88 + 12 * K),
        'deform_backward': P * (40 + 16 * K),
    }.get(name)


def cpu_baseline_sp(cfg, M, K, F, seconds_budget=12.0):
    """the stage on the host cores: the oracle's search (3 + F dimensions over M superpoints), weighting, skinning and
    rasterizer forward + backward (OpenMP), and the torch CPU restatement of sp_deform_net forward + backward; bounded sample"""
    import time as _t
    import numpy as np
    from benchlib.cpu_baseline import _cpu_oracle
    from sk_gs_amd import scene
    from sk_gs_amd.superpoint import SpDeformNet
    o, lib = _cpu_oracle()
    P, W, H = cfg['P'], cfg['W'], cfg['H']
    g = scene.make_gaussians(P, seed=0)
    cam = scene.make_camera(W, H, seed=0)
    rs = scene.raster_settings_from_camera(cam, colmap=True)
    n = lambda t: t.numpy()  # noqa: E731
    gen = torch.Generator().manual_seed(5)
    sp = g['xyz'][torch.randperm(P, generator=gen)[:M]].clone()
    feat, sfeat = 0.02 * torch.randn(P, F, generator=gen), 0.02 * torch.randn(M, F, generator=gen)
    pts, sps = np.concatenate([n(g['xyz']), n(feat)], 1), np.concatenate([n(sp), n(sfeat)], 1)
    radius, kw = np.full(M, 0.26), np.full(M, 0.5)
    torch.manual_seed(0)
    net = SpDeformNet()
    t = torch.tensor([0.3])
    gcol, gop = torch.randn(3, H, W, generator=gen).numpy(), torch.randn(H, W, generator=gen).numpy()

    def one():
        out = net.reference_forward(sp, t)
        q = torch.nn.functional.normalize(out['d_rotation'] + torch.tensor([0, 0, 0, 1.]), dim=-1)
        spT = torch.cat([out['d_xyz'], q], 1)
        dist, idx = o.knn_bones(pts, sps, K)
        w = o.lbs_weights_kernel(dist, idx, radius, kw)
        d = o.lbs_deform_forward(n(g['xyz']), w, idx, n(spT.detach()), n(q.detach()), n(out['d_scaling'].detach()), n(g['xyz']),
                                 n(g['log_scale']), n(g['rot']), n(g['opacity_logit']))
        fwd = o.rasterize_forward(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix), n(rs.projmatrix),
                                  n(rs.campos), d['means'], d['opacity'], n(g['sh']), d['scales'], d['rotations'])
        gr = o.rasterize_backward(fwd, H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix), n(rs.projmatrix),
                                  n(rs.campos), d['means'], n(g['sh']), d['scales'], d['rotations'], gcol, gop)
        gb = o.lbs_deform_backward(n(g['xyz']), w, idx, n(spT.detach()), n(q.detach()), n(out['d_scaling'].detach()),
                                   n(g['log_scale']), n(g['rot']), n(g['opacity_logit']), gr['dL_dmeans3D'], gr['dL_dscales'],
                                   gr['dL_drotations'], gr['dL_dopacity'])
        (spT * torch.from_numpy(np.asarray(gb['g_bone_T']))).sum().backward()
        net.zero_grad()

    one()
    t0, it = _t.perf_counter(), 0
    while True:
        one()
        it += 1
        if _t.perf_counter() - t0 > seconds_budget or it >= 30:
            break
    el = _t.perf_counter() - t0
    return dict(value=round(it / el, 4), unit='iters/s', cores=o.num_threads(), kind='port',
                sample=f'{it} iterations of stage sp on the host: sp_deform_net (torch CPU) + the oracle\'s 3+{F}-d search over {M} '
                       f'superpoints, weighting, skinning, rasterize forward + backward, skinning backward (no loss / Adam), '
                       f'{el:.1f} s, oracle built {"-march=native" if lib else "portable"}, OpenMP')


def run(args, base_alg_bytes, configs):
    """`args`: bench.py's namespace (config, views, steps, warmup, lr, lbs_method ...)"""
    from sk_gs_amd import _C, scene
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.superpoint import FusedSuperpointStep, FusedSuperpointTrainStep, SuperpointGaussians
    from sk_gs_amd.train_step import GraphedSteps
    from sk_gs_amd.view_parallel import ViewParallel, init_distributed
    from sk_gs_amd.view_slot import ViewTable

    rank, world, local_rank = init_distributed()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree'
    use_dist = dist.is_initialized()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(local_rank)
    cfg = configs[args.config]
    P, W, H = cfg['P'], cfg['W'], cfg['H']
    if args.preset == 'sc_gs':  # exps/d_nerf_sc_gs.yaml:18,31-32 (sep_rot: the class default, not overridden there)
        args.lbs_method, args.warp_method, args.sep_rot, args.knn = 'weighted_kernel', 'LBS_c', True, 3
    if args.preset == 'sp_gs':  # exps/d_nerf_sp_gs.yaml:18,30-32
        args.lbs_method, args.warp_method, args.sep_rot, args.knn = 'W', 'largest', False, 3
    M, K, F = args.superpoints, args.knn, 8
    frames = args.views
    model = SuperpointGaussians(P, M, K, sh_degree=3, num_frames=frames, seed=0, scale_mult=args.scale_mult,
                                lbs_method=args.lbs_method, hyper_dim=F, warp_method=args.warp_method, sep_rot=args.sep_rot,
                                is_blender=not args.raw_time, t_degree=10 if args.raw_time else 6).to(dev)
    model.time_noise = 0.01 if args.raw_time else 0.0  # (time_interval 1/100 x smooth scale 1: the start of the annealing)
    if not args.keep_order:  # Gaussians along a Z-order curve (sk_gs_amd/densify.py::sort_spatially): what a training loop
        from sk_gs_amd.densify import sort_spatially  # does after every densification event
        sort_spatially(model)
    cams = [scene.make_camera(W, H, seed=i) for i in range(args.views)]
    settings = [scene.raster_settings_from_camera(c, sh_degree=3, colmap=True, device=dev) for c in cams]
    background = torch.ones(3, device=dev)
    gen = torch.Generator().manual_seed(77)
    _C.config.sync_num_rendered = True
    targets, Rs, longest, walked = [], [], 0, []
    with torch.no_grad():
        for v in range(args.views):
            out = model.render(settings[v], time_id=v % frames, background=background)
            targets.append((out['images'] + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1).contiguous())
            buf = out['buffer']
            Rs.append(buf.R)
            longest = max(longest, _C.read_status(buf.geomBuffer)['max_tile_count'])
            walked.append(int(buf.imgBuffer[:W * H * 4].view(torch.int32).sum(dtype=torch.int64)))
    R_mean, R_max = sum(Rs) / len(Rs), max(Rs)
    tile_bucket = ((int(longest * 1.5) + 63) // 64) * 64
    if 512 < tile_bucket and longest * 1.2 <= 512:
        tile_bucket = 512
    _C.config.sync_num_rendered = False
    view_table = ViewTable(settings, [float(model.frame_times[v % frames]) for v in range(args.views)],
                           [v % frames for v in range(args.views)], torch.stack(targets), dev)
    vp = ViewParallel(model.parameters(), average=True)  # the parameters' gradients as views of ONE flat buffer
    step = FusedSuperpointStep(model, W, H, capacity=int(R_max * 1.25 * _C.config.capacity_growth) + 1024,
                               background=background, grad_scale=1.0 / world, tile_bucket=tile_bucket, view_table=view_table)
    opt = FusedAdam(model.param_groups(lr=args.lr), eps=1e-15, betas=(0.9, 0.999))
    from benchlib.sk_stage import lr_schedules
    lr_schedule_desc = lr_schedules(args, opt)  # update_learning_rate on the device: xyz and sp_deform (sk_gs.py:627-631)
    train = FusedSuperpointTrainStep(step, opt, enable=not use_dist)
    order = [vp.view_index(i, args.views) for i in range(args.views)]
    train_chunk, n_multi = None, 1
    if not use_dist:
        view_table.set_order(order)
        g_step = GraphedSteps(lambda _: train(), collect_garbage=False)

        def train_step(i):
            g_step(0)
        if args.steps_per_graph > 1:  # several consecutive steps per replay (the closing launch selects the next view)
            n_multi = int(args.steps_per_graph)

            def train_chunk(i, n):
                g_step.replay(0, n)
    else:  # view-parallel ranks: backward | ONE all-reduce of the flat gradient buffer | Adam
        g_bwd = GraphedSteps(lambda _: step.forward_backward(), collect_garbage=False)
        g_opt = GraphedSteps(lambda _: opt.step(), collect_garbage=False)

        def train_step(i):
            view_table.select(vp.view_index(i, args.views))
            g_bwd(0)
            vp.allreduce_grads(prescaled=True)
            g_opt(0)

    def eager_step(i):
        if not use_dist:
            train()
        else:
            view_table.select(vp.view_index(i, args.views))
            step.forward_backward()
            vp.allreduce_grads(prescaled=True)
            opt.step()

    eager_step(0)
    if not use_dist:
        view_table.rewind()
    if train_chunk is not None:  # (the one-step graph first: a first call captures)
        train_step(0)
        g_step.capture(0, repeat=n_multi, warmup=0)
        view_table.rewind()
    if args.prime_steps > 0:  # setup: the steady state of a training run before anything is timed (see --prime-steps)
        i = 0
        while i < args.prime_steps:
            if train_chunk is not None and args.prime_steps - i >= n_multi:
                train_chunk(i, n_multi)
                i += n_multi
            else:
                train_step(i)
                i += 1
        torch.cuda.synchronize()
        if not use_dist:
            view_table.rewind()
    for i in range(max(args.warmup, 2)):
        train_step(i)
    elapsed, block_stats = timing.timed_steps(train_step, args.steps, args.warmup, dev, chunk=n_multi, train_chunk=train_chunk)
    replicas_identical, param_digest = timing.replicas_digest(model, world) if use_dist else (None, None)
    st = step.status()
    assert st['overflow_events'] == 0, 'binning capacity overflow during the timed region: result invalid'
    assert st['pairs_overflow_events'] == 0, 'a superpoint\'s inverse neighbour list overflowed in some step: result invalid'
    # ---- per-kernel HIP-event timing: an eager pass over the same steps
    n_prof = min(args.steps, 20)
    prof = timing.profiled_eager_pass(_C, eager_step, args.warmup + args.steps, n_prof)
    rows_b = 28 * sum(p.numel() for g in opt.param_groups if g.get('name') in train.rows for p in g['params'])
    rest_b = 28 * sum(p.numel() for g in opt.param_groups if g.get('name') in train.rest + train.wide for p in g['params'])
    kernels = {}
    for name, (ms, n) in prof.items():
        us = ms / n * 1e3
        b = alg_bytes_sp(name, P, M, K, F, W, H, R_mean, rows_adam_bytes=rows_b if train.fused else 0)
        if b is None:
            b = base_alg_bytes(name, P, M, K, W, H, R_mean)
        if name == 'adam':
            b = rest_b if train.fused else rows_b + rest_b
        # the skinning and the rows pass of its backward run as jobs of the rasterizer's per-Gaussian launches
        # (skgs_raster_inputs.deform_job / skgs_raster_grads.sp_skinning_job): their bytes, minus the 44 B per Gaussian of
        # means / scales / rotations / opacity (and their gradients) that are no longer re-read
        includes = None
        if name == 'preprocess_forward' and 'deform_forward' not in prof and step.deform_in_preprocess:
            b += alg_bytes_sp('deform_forward', P, M, K, F, W, H, R_mean) - 44 * P
            includes = 'deform_forward (skinning + activations)'
        if name == 'preprocess_backward' and step.deform_backward_in_preprocess:
            b += P * (40 + 16 * K) - 44 * P
            includes = 'the rows pass of deform_backward (the row of that name: its bones + finalize launches)'
        if name == 'deform_backward' and step.deform_backward_in_preprocess:
            b = P * 16 * K + M * 24 * 4  # the bones pass gathers the pairs' payload; per-superpoint sums out
        rec = timing.kernel_record(us, n / n_prof, b)
        if includes:
            rec['includes'] = includes
        if name in ('sp_net_forward', 'sp_net_backward'):
            fl = net_flops(M)[0 if name == 'sp_net_forward' else 1]
            rec['TFLOPs'] = round(fl / (us * 1e-6) / 1e12, 2)
            rec['frac_of_mfma_f32_peak'] = round(fl / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        kernels[name] = rec
    # forward-only render rate of the stage (the reference's FPS protocol, test.py:102-123: warm-up, then N renders between two
    # events, views cycled): sp net + search + skinning + rasterize as one graph replay per render
    fps = None
    if world == 1:
        if not use_dist:
            view_table.clear_order()
        g_fwd = GraphedSteps(lambda _: step.forward(), collect_garbage=False)
        view_table.select(0)
        g_fwd.capture(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(220):
            if i == 20:
                e0.record()
            view_table.select(i % args.views)
            g_fwd(0)
        e1.record()
        torch.cuda.synchronize()
        fps = dict(fused_step_graph=round(200 * 1000.0 / e0.elapsed_time(e1), 1),
                   protocol='test.py:102-123: 20 warm-up + 200 renders between two events, views cycled')
    from benchlib import launch as _launch
    cluster = _launch.cluster_info(dist, torch, local_rank)  # (a collective: every rank)
    if rank != 0:
        return None
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_sp(cfg, M, K, F, min(args.cpu_seconds, 12.0))
        except Exception as e:  # noqa
            cpu = dict(error=f'{type(e).__name__}: {e}'[:300])
    rb = kernels.get('render_backward', {})
    rb_us, rb_bytes = rb.get('us', 0.0), base_alg_bytes('render_backward', P, M, K, W, H, R_mean)
    achieved = rb_bytes / (rb_us * 1e-6) / 1e9 if rb_us else 0.0
    T = ((W + 15) // 16) * ((H + 15) // 16)
    step_bytes = (P * (1138 + 4 * K) + 116 * R_mean + 44 * W * H + 8 * T  # SURVEY 8d without the 4 M logit row (kernel weighting)
                  + alg_bytes_sp('sp_knn_weights', P, M, K, F, W, H, R_mean) + alg_bytes_sp('sp_knn_weights_backward', P, M, K, F, W, H, R_mean)
                  + alg_bytes_sp('sp_net_forward', P, M, K, F, W, H, R_mean) + alg_bytes_sp('sp_net_backward', P, M, K, F, W, H, R_mean))
    ms_step = elapsed / args.steps * 1e3
    from benchlib.roofline import committed_counter_profile
    from_profile = committed_counter_profile(cfg['name'] + ', stage sp', 'pmc_render_backward_sp.json')
    return {
        'metric': f'train iters/sec, SUPERPOINT stage (sp net + 3+8-d search + skinning + rasterize fwd+bwd + L1/SSIM loss + Adam; image '
                  f'loss only: the shipped `sparse` / `smooth` weight regularisers are not included), {P // 1000}k Gaussians @{W}x{H}',
        'value': round(world * args.steps / elapsed, 3), 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'prime_steps': args.prime_steps, 'steps_per_graph': n_multi if train_chunk is not None else 1,
        'ms_per_step': round(ms_step, 4), 'ms_per_step_blocks': block_stats, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{cfg["name"]} in stage sp: {P} Gaussians, {M} superpoints, K={K}, search over xyz + {F} hyper '
                               f'dimensions, LBS_method {args.lbs_method}, warp_method {args.warp_method}, sep_rot {bool(args.sep_rot)}, '
                               f'sp_deform_net 8x256 on {M} rows' + (' (is_blender=False: raw time encoding of degree 10, time noise 0.01)' if args.raw_time else '') + f', SH degree 3, {W}x{H}, '
                               f'{args.views} synthetic views, 1 view per rank per step',
                   'stage': 'sp',
                   'loss': 'image terms only (0.8 L1 + 0.2 (1 - SSIM)): the shipped sp configuration also runs two regularisers on the [P,K] LBS '
                           'weights in every iteration (sparse, smooth, weight 0.1 each: exps/default.yaml:85-86, sk_gs.py:1339-1359,1572-1574) '
                           '-- the reference\'s own torch code, SURVEY 2 out of scope; `--stage sp --reference-loop fused --sp-regularisers` times '
                           'the iteration with them',
                   'num_rendered_mean': round(R_mean), 'num_rendered_max': R_max,
                   'tile_list_mean': round(R_mean / T, 1), 'tile_list_max': longest,
                   'walked_pairs_mean': round(sum(walked) / len(walked)),
                   'parallelism': f'view-parallel x{world}, flat-buffer grad all-reduce ({vp.grads.nbytes / 1e6:.1f} MB)',
                   'launch': 'ONE captured hipGraph for all views (camera, time and target read from a device view slot)'
                   if not use_dist else 'two hipGraphs per step with the all-reduce between them',
                   'tile_lists': f'buckets of {tile_bucket} slots per tile (longest list {longest})',
                   'lr': args.lr, 'lr_schedule': lr_schedule_desc,
                   'adam': ('per-Gaussian rows on the idle CUs of the sp net\'s two backward launches; '
                            + ('the dense [P, M] logit table as a launch of its own; ' if train.wide else '')
                            + 'network + superpoint tables + counter + next view in one closing launch') if train.fused
                   else 'one launch after the all-reduce',
                   'step': 'FusedSuperpointStep (direct C-ABI calls)', 'cluster': cluster,
                   'exchange': 'allreduce' if world > 1 else None, 'replicas_identical': replicas_identical, 'param_digest': param_digest,
                   'gaussian_order': 'as generated (random)' if args.keep_order else
                   'sorted along a Z-order curve (densify.sort_spatially: what a training loop does after each densification event)'},
        'cpu_baseline': cpu, 'fps_forward_render': fps,
        'roofline': {'bound': 'hbm', 'kernel': 'render_backward', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBPS,
                     'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 5),
                     'traffic': (from_profile or {}).get('hbm_bytes_per_launch'),
                     'traffic_source': ({k: from_profile.get(k) for k in ('file', 'commit')} if from_profile else None), 'avg_us': rb_us,
                     'alg_bytes_per_launch': int(rb_bytes), 'limiter': 'valu',
                     'whole_step': {'alg_bytes': int(step_bytes), 'ms': round(ms_step, 4),
                                    'frac': round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
                     'note': 'same dominant kernel as stage sk (VALU-issue bound); the stage\'s own kernels are under "kernels": '
                             'the sp net against the fp32 MFMA peak, the search against its algorithmic bytes'},
        'kernels': kernels,
    }
