"""The timed region of bench.py and the per-kernel table behind it (both stages)."""
import time

import torch
import torch.distributed as dist

from benchlib.options import HBM_PEAK_GBPS


def timed_steps(train_step, steps, warmup, dev, after_step=None):
    """exactly `steps` calls of train_step(warmup + i) between a barrier + synchronize on both sides.  Returns (seconds: MAX over
    the ranks, block statistics of rank 0's launch stream).  An event every steps/10 steps splits the region into >= 10 blocks
    (when steps >= 10) whose per-step times give the spread of `ms_per_step` (median / p10 / p90); `after_step(i)` runs
    inside the region (the densification events of --densify-every)."""
    use_dist = dist.is_initialized()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    n_blocks = min(steps, 10)
    edges = [round(b * steps / n_blocks) for b in range(n_blocks + 1)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in edges]
    t0 = time.perf_counter()
    marks[0].record()
    nxt = 1
    for i in range(steps):
        train_step(warmup + i)
        if after_step is not None:
            after_step(i)
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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, block_stats


def kernel_record(us, launches_per_step, alg_b):
    """one entry of the JSON line's `kernels` table: HIP-event time per launch, algorithmic bytes, GB/s, fraction of the HBM peak"""
    return dict(us=round(us, 2), launches_per_step=round(launches_per_step, 2),
                alg_MB=round(alg_b / 1e6, 2) if alg_b else None,
                GBps=round(alg_b / (us * 1e-6) / 1e9, 1) if alg_b else None,
                frac=round(alg_b / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if alg_b else None)


def profiled_eager_pass(_C, eager_step, first, count):
    """per-kernel HIP-event timing: an eager pass over the same steps (events cannot be read back from inside a replayed graph;
    the kernels and their inputs are the same).  Returns {kernel: (ms total, launches)}"""
    _C.profile_enable(None)
    for i in range(count):
        eager_step(first + i)
    torch.cuda.synchronize()
    prof = _C.profile_collect()
    _C.profile_enable([])
    return prof


def replicas_digest(model, world):
    """view-parallel replicas must stay bit-identical: same reduced gradients, same Adam step on every rank.  Returns
    (True | names of the parameters that differ, sum |p| over all parameters on this rank: compares exchange modes)"""
    names = [n for n, _ in model.named_parameters()]
    digest = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                         [p.detach().double().abs().sum() for p in model.parameters()])
    every = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(every, digest)
    differ = sorted({names[i % len(names)] for e in every for i in (every[0] != e).nonzero().flatten().tolist()})
    return (True if not differ else differ), float(digest[len(names):].sum())
