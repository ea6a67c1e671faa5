"""bench.py's tables and command line: the BASELINE workloads, SURVEY 8(d)'s algorithmic bytes per kernel, the exchange variants
of a multi-rank run and the argument parser."""
import argparse

CONFIGS = {
    0: dict(name='static-10k-400', P=10_000, M=0, K=0, W=400, H=400),
    1: dict(name='hook-like-100k-800', P=100_000, M=20, K=5, W=800, H=800),
    2: dict(name='atlas-like-200k-512', P=200_000, M=32, K=5, W=512, H=512),
    3: dict(name='mutant-like-300k-800', P=300_000, M=20, K=5, W=800, H=800),
    4: dict(name='zju-like-500k-1024', P=500_000, M=24, K=5, W=1024, H=1024),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

# world > 1: how the gradients cross the wire.  The order is the order `--exchange auto` times them in: the
# predicted-best variant with EAGER collectives first (`factors` 5.9x, `factors-overlap` 6.0x against 4.7x for the plain
# all-reduce), so a caller's time limit that cuts the ranking short still records it; the plain all-reduce second; the
# captured-collective variants last, each under the watchdog (a fabric on which a captured collective never completes costs
# only those entries)
EXCHANGE_VARIANTS = {
    'allreduce': dict(),
    'factors': dict(sh_factors=True, compact_logits=True),
    'factors-overlap': dict(sh_factors=True, compact_logits=True, overlap_gather=True),
    'pipeline': dict(pipeline=True),
    'allreduce-graph': dict(graph_collectives=True),
    'factors-graph': dict(sh_factors=True, compact_logits=True, overlap_gather=True, graph_collectives=True),
    'factors-graph-split': dict(sh_factors=True, compact_logits=True, overlap_gather=True, graph_collectives=True, split_rest=True),
}


# the launches of ONE fused training step of stage sk on one rank, as rocprofv3 names them (regex on the kernel name, template
# arguments included where the step's instantiation differs from the operator path's): tools/pmc_summary.py sums their average
# durations out of the committed `--kernel-trace --stats` pass; bench.py prints that sum beside its own event-bracketed table
STEP_KERNELS_SK = {
    'skeleton_forward': r'fused_mlp_forward_kernel<2>', 'preprocess_forward': r'preprocess_forward_kernel<true, [1-9]\d*>',
    'scatter': r'scatter_lds_kernel<4>', 'tile_sort': r'tile_sort_wave_kernel', 'render_forward': r'render_forward_kernel<1, 0, false>',
    'image_loss_forward': r'image_loss_forward_kernel', 'image_loss_backward': r'image_loss_backward_kernel',
    'render_backward': r'render_backward_kernel<1, 0>', 'preprocess_backward': r'preprocess_backward_kernel<true, [1-9]\d*>',
    'deform_backward_finalize': r'deform_backward_finalize_kernel', 'skeleton_backward': r'fused_mlp_backward_kernel<2>',
    'adam': r'adam_step_kernel',
}


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


def whole_step_bytes(P, M, K, W, H, R):
    """B_alg of SURVEY 8(d): one render's algorithmic bytes, forward + backward"""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return P * (1138 + 4 * M + 4 * K) + 116 * R + 44 * W * H + 8 * T


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--steps-per-graph', type=int, default=4,
                    help='consecutive training steps recorded in ONE hipGraph (one-rank fused step with an ordered view table, whose '
                         'closing launch selects the next view; the one-step graph serves the remainder): between two replays '
                         'the device idles ~8 us.  1 = one step per replay')
    ap.add_argument('--prime-steps', type=int, default=100,
                    help='untimed steps run as part of the SETUP (after the graph capture, before the --warmup steps): a fresh '
                         'process reaches the steady state of a training run -- clocks, TLBs, the views\' first touches -- only '
                         'after some tens of ms of work; with the driver\'s 5 warm-up steps (1.9 ms) a 20-step region measured '
                         '0.375 ms per step, 0.358 behind 50 and the long-run 0.354 behind 200.  Reported as `prime_steps`; 0 = off')
    ap.add_argument('--config', type=int, default=1)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--ppl', type=int, default=0, help='pixels per lane of the blend kernels (0 = heuristic)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--list-headroom', type=float, default=1.0,
                    help='multiplies the room the tile lists and the binning capacity get over what the first renders needed (a drifting '
                         'workload -- U(0,1) targets -- outgrows the default 1.2-1.5x; overflow inside the timed region fails the run)')
    ap.add_argument('--no-survey-recipe', action='store_true',
                    help="skip the second short run the default one-GPU line carries under `survey_recipe`: SURVEY.md 8(d)'s recipe "
                         "to the letter (U(0,1) targets, Gaussians in generation order) in a child process")
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--cpu-single-thread', action='store_true',
                    help='also time the oracle on the bench workload with ONE thread (about a minute per iteration at config #1)')
    ap.add_argument('--ms-per-render', action='store_true', default=None,
                    help='also time rasterizer fwd+bwd alone (operator path, fixed upstream gradients, host-synchronised '
                         'per render: median / p10 / p90 of 50); default: on for a 1-GPU run')
    ap.add_argument('--no-ms-per-render', dest='ms_per_render', action='store_false')
    ap.add_argument('--lr', type=float, default=1e-4, help='base lr (reference 1e-3); small keeps the workload stationary')
    ap.add_argument('--lr-schedule', choices=('device', 'off'), default='device',
                    help="the reference's per-iteration update_learning_rate (train.py:140-141): `xyz` follows get_expon_lr_func from "
                         "0.16 lr to 0.0016 lr over 30 000 steps (gaussian_splatting.py:455-470), the deform network's group its own "
                         "schedule over 40 000 (sk_gs.py:611-632) -- evaluated ON THE DEVICE by the step's closing Adam launch "
                         "(FusedAdam.set_lr_schedule), so every step of a multi-step graph replay applies its own rate.  off: constant rates")
    ap.add_argument('--targets', choices=('own-render', 'uniform'), default='own-render',
                    help="training targets: the model's own initial renders + N(0, 0.05^2) noise (default: gradients stay small, the "
                         "workload -- num_rendered, tile lists -- is stationary over the run) or U(0,1) images, SURVEY.md 8(d)'s recipe")
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
    ap.add_argument('--exchange', choices=('auto',) + tuple(EXCHANGE_VARIANTS), default='auto',
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
    ap.add_argument('--autograd-fused', action='store_true',
                    help="the reference's loop shape over the FUSED launches: `loss = step.loss(...); loss.backward(); optimizer.step()` "
                         "(FusedTrainStep.loss: one autograd node whose backward carries the per-Gaussian rows' update on its skeleton launch; "
                         "optimizer.step() is then the closing launch.  With --serial-adam: FusedViewStep.loss and a full optimizer launch)")
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
    ap.add_argument('--no-reference-route', action='store_true',
                    help="default line: skip the three short child runs of `--reference-loop` (per-method fast paths; the fused route in stage sk and "
                         "stage sp) reported under `reference_route`")
    ap.add_argument('--loop-scene', choices=('r5', 'headline'), default='r5',
                    help="--reference-loop, stage sk: 'r5' = the synthetic scene of round 5's reference-loop lines (head weights of the "
                         "deform network N(0, s/16): joint rotations of ~0.2 rad along the chain pile the Gaussians up -- R = 0.8 M tile "
                         "instances, lists of up to 770); 'headline' = the default bench line's scene (heads x 0.01: rotations near identity, "
                         "Gaussians in Z-order, R = 0.52 M)")
    ap.add_argument('--sp-regularisers', action='store_true',
                    help="--stage sp --reference-loop accelerated|fused: add the two per-Gaussian regularisers the shipped config runs on the "
                         "[P,K] LBS weights in every sp iteration (`sparse`, `smooth`, weight 0.1 each: exps/default.yaml:85-86, "
                         "sk_gs.py:1339-1359,1572-1574) as the reference writes them (torch); without the flag the loss is the image "
                         "terms only")
    ap.add_argument('--reg-torch', action='store_true',
                    help='--sp-regularisers: the two regularisers as the reference writes them in torch (a [P, 21, K] gather and its sort-based '
                         'index backward) instead of the one-launch kernels accelerate_reference() patches in (csrc/weight_reg.hip)')
    ap.add_argument('--superpoints', type=int, default=512, help='--stage sp: num_superpoints (exps/default.yaml:25)')
    ap.add_argument('--lbs-method', choices=('weighted_kernel', 'kernel', 'dist', 'W'), default='weighted_kernel',
                    help="--stage sp: the weighting of calc_LBS_weight (class default 'weighted_kernel', sk_gs.py:364; "
                         "exps/default.yaml:35 sets 'W': a dense [P,512] logit table)")
    ap.add_argument('--warp-method', choices=('LBS', 'LBS_c', 'largest'), default='LBS',
                    help="--stage sp: warp_method (exps/default.yaml:36 LBS; exps/d_nerf_sc_gs.yaml:32 LBS_c: the superpoint's transform "
                         "re-centred on the superpoint, sk_gs.py:803-804)")
    ap.add_argument('--sep-rot', action='store_true',
                    help="--stage sp: the deform network's local_rotation head feeds the rotation blend (sep_rot, the class default "
                         "sk_gs.py:357 and what exps/d_nerf_sc_gs.yaml runs with)")
    ap.add_argument('--reference-loop', choices=('hooks', 'accelerated', 'fused'), default=None,
                    help="time the REFERENCE's own call sequence (benchlib/ref_sequence.py + train.py's step) on install_reference_hooks() "
                         "alone, after accelerate_reference(fused_render=False) (per-method fast paths), or -- fused -- with "
                         "SkeletonGaussianSplatting.render / ImageLoss.forward / SSIM_Loss.forward routed into the fused step "
                         "(sk_gs_amd/reference_fused.py): what a user of the unmodified reference gets (benchlib/reference_loop.py)")
    ap.add_argument('--raw-time', action='store_true',
                    help="--stage sp: DeformNetwork(is_blender=False) (sk_gs.py:220,255-261; no shipped YAML): no time network, time "
                         "degree 10, and the stage's time noise (sk_gs.py:837-839) drawn on the device inside the captured step")
    ap.add_argument('--knn', type=int, default=5, help='--stage sp: num_knn (exps/default.yaml:20: 5; the SC-GS / SP-GS configs: 3)')
    ap.add_argument('--preset', choices=('sc_gs', 'sp_gs'), default=None,
                    help="--stage sp: sc_gs = exps/d_nerf_sc_gs.yaml's combination: LBS_method weighted_kernel, warp_method LBS_c, "
                         "sep_rot, num_knn 3")
    return ap


def exchange_flags_given(args) -> bool:
    return bool(args.pipeline or args.compact_logits or args.sh_factors or args.overlap_gather or args.graph_collectives
                or args.split_rest or args.exchange != 'auto')


def apply_exchange(args, name):
    for k, v in EXCHANGE_VARIANTS[name].items():
        setattr(args, k, v)


def exchange_name(args, world):
    """the name of the exchange the flags in `args` describe (None on one rank)"""
    if world <= 1:
        return None
    base = ('pipeline' if args.pipeline else 'factors-overlap' if args.overlap_gather else 'factors' if args.sh_factors else
            'compact-logits' if args.compact_logits else 'allreduce')
    return base + ('-graph' if args.graph_collectives else '') + ('-split' if args.split_rest else '')
