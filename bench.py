#!/usr/bin/env python3
"""Benchmark of the SK_GS hot path on MI355X: train iters/s (deform -> rasterize -> 0.8 L1 + 0.2 (1-SSIM) -> backward ->
[all-reduce] -> Adam(eps=1e-15)) for BASELINE.json config #1: 100k Gaussians, 20 bones, K=5, SH degree 3, 800x800.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  A "step" is one training iteration of ONE view per rank (the reference trains with
batch = 1 view, data_loader/build.py:26); with N ranks N views are processed per step (weak scaling) and `value` is
the whole-job rate N*K / t.  Inputs (Gaussians, cameras, bones, target images) are synthetic, seeded and resident in
HBM before the timed region.  Extra objects on the line:
  roofline      dominant kernel (render_backward): algorithmic bytes (40 R + 24 W H + 44 P) / HIP-event time,
                the committed counter profile's HBM traffic, the whole step against the same roof
  kernels       per-kernel HIP-event time, algorithmic GB/s and fraction of the HBM peak (separate short pass after the timed region)
  cpu_baseline  the CPU oracle (restatement of the reference kernels; the reference has no CPU path) on the host cores

The pieces live in benchlib/: options.py (workload tables, command line), launch.py (`--gpus N` without a launcher),
sk_stage.py / sp_stage.py (one measurement of a stage), timing.py (the timed region), render_protocol.py (ms/render, FPS),
roofline.py, exchange_rank.py (`--exchange auto`), cpu_baseline.py.
"""
import json
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from benchlib import launch, options  # noqa: E402
from benchlib.options import CONFIGS, alg_bytes  # noqa: E402,F401  (the tables, under their old names)


def claim_stdout():
    """stdout carries exactly ONE JSON line: everything else that any library prints to fd 1 (RCCL's version banner, MIOpen
    notes) is diverted to stderr for the whole run.  Returns write_line(dict)"""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def write_line(line):
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    return write_line


def main(argv=None):
    args = options.build_parser().parse_args(argv)
    if launch.needs_spawn(args.gpus):  # `python bench.py --gpus N` without a launcher: start the N ranks (or refuse), never run 1
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else argv))
    import sk_gs_amd
    from sk_gs_amd import _C
    from sk_gs_amd.view_parallel import init_distributed
    sk_gs_amd.single_thread_backward(args.backward_thread == 'caller')
    write_line = claim_stdout()
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'

    if args.reference_loop:
        from benchlib import reference_loop
        write_line(reference_loop.run(args, CONFIGS))
    elif args.stage == 'sp':
        from benchlib import sp_stage
        line = sp_stage.run(args, alg_bytes, CONFIGS)
        if line is not None:
            write_line(line)
    else:
        # SKGS_FORCE_DIST=1: create a (1-rank) RCCL group so a single GPU exercises the multi-GPU code path
        rank, world, local_rank = init_distributed(force=bool(os.environ.get('SKGS_FORCE_DIST')))
        assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree'
        torch.cuda.set_device(local_rank)
        env = SimpleNamespace(rank=rank, world=world, local_rank=local_rank, dev=torch.device('cuda', local_rank),
                              use_dist=dist.is_initialized())
        _C.load_library()
        from benchlib import sk_stage
        explicit = options.exchange_flags_given(args)
        if args.exchange != 'auto':
            options.apply_exchange(args, args.exchange)
        if world > 1 and not explicit and not args.autograd and not args.torch_adam:
            from benchlib import exchange_rank
            exchange_rank.rank_variants(args, lambda a: sk_stage.run(a, env), rank, env.dev, write_line)
        else:
            line = sk_stage.run(args, env)
            if rank == 0:
                line['config']['exchange'] = options.exchange_name(args, world)
                write_line(line)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
