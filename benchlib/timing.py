"""The timed region of bench.py and the per-kernel table behind it (both stages)."""
import time

import torch
import torch.distributed as dist

from benchlib.options import HBM_PEAK_GBPS


def timed_steps(train_step, steps, warmup, dev, after_step=None, chunk=1, train_chunk=None):
    """exactly `steps` training steps -- train_step(warmup + i), or `train_chunk(warmup + i, chunk)` = `chunk` consecutive steps in
    one graph replay while that many are left -- between a barrier + synchronize on both sides.  Returns (seconds: MAX over
    the ranks, block statistics of rank 0's launch stream).  Events split the region into ~10 blocks (when steps >= 10) whose
    per-step times give the spread of `ms_per_step` (median / p10 / p90); `after_step(i)` runs inside the region behind step i
    (the densification events of --densify-every; the caller keeps `chunk` a divisor of their period)."""
    use_dist = dist.is_initialized()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if train_chunk is None:
        chunk = 1
    n_blocks = max(1, min(steps, 10))
    edges = [round(b * steps / n_blocks) for b in range(1, n_blocks + 1)]
    marks = [(torch.cuda.Event(enable_timing=True), 0)]
    t0 = time.perf_counter()
    marks[0][0].record()
    i = 0
    while i < steps:
        n = chunk if steps - i >= chunk else 1
        if n > 1:
            train_chunk(warmup + i, n)
        else:
            train_step(warmup + i)
        if after_step is not None:
            for j in range(i, i + n):
                after_step(j)
        i += n
        if edges and i >= edges[0]:
            while edges and i >= edges[0]:
                edges.pop(0)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((e, i))
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[b][0].elapsed_time(marks[b + 1][0]) / (marks[b + 1][1] - marks[b][1]) for b in range(len(marks) - 1))
    nb = len(per_step)
    block_stats = dict(blocks=nb, median=round(per_step[nb // 2], 4), p10=round(per_step[nb // 10], 4),
                       p90=round(per_step[min(nb - 1, (9 * nb) // 10)], 4),
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
