"""`bench.py --exchange auto` on more than one rank: time EVERY exchange variant and report the fastest identical one.

A multi-rank run with no exchange flag times every variant of benchlib.options.EXCHANGE_VARIANTS -- same scene, same seed, the
runtime rebuilt for each, exactly args.steps steps between barriers each -- and reports the FASTEST ONE WHOSE REPLICAS STAYED
IDENTICAL as `value`; all of them are listed under `exchange_variants`.  No multi-GPU box exists in the build loop: the ranking is
done where the xGMI links are (DESIGN.md section 6 holds the predicted table to read the record against).
"""
import copy
import gc
import os
import threading
import time

import torch
import torch.distributed as dist

from benchlib.options import EXCHANGE_VARIANTS


def summarise(lines, errors, aborted=None):
    """(best variant name, per-variant summary) from rank 0's lines"""
    summary = {}
    for name in EXCHANGE_VARIANTS:
        ln = lines.get(name)
        if ln is None:
            summary[name] = dict(error=errors.get(name, 'abandoned: it did not finish in time' if name == aborted else 'not run'))
        else:
            summary[name] = dict(value=ln['value'], ms_per_step=ln['ms_per_step'], ms_per_step_blocks=ln.get('ms_per_step_blocks'),
                                 parallelism=ln['config']['parallelism'], replicas_identical=ln['config']['replicas_identical'],
                                 param_digest=ln['config']['param_digest'])
    good = [n for n in EXCHANGE_VARIANTS if lines.get(n) is not None and lines[n]['config']['replicas_identical'] is True]
    assert good, f'no exchange variant kept the replicas identical: {summary}'
    return max(good, key=lambda n: lines[n]['value']), summary


def rank_variants(args, run_workload, rank, dev, write_line):
    """`run_workload(args)` -> the JSON line as a dict on rank 0 (None elsewhere); `write_line(dict)` prints it (rank 0)"""
    lines, errors = {}, {}

    def emit(aborted=None):
        if rank != 0:
            return
        best, summary = summarise(lines, errors, aborted)
        line = lines[best]
        line['config']['exchange'] = best
        line['exchange_variants'] = summary
        write_line(line)

    def provisional(name):
        # N-rank evidence must not wait for the whole ranking.  As soon as a variant has finished -- the FIRST is the
        # plain flat-buffer all-reduce, nothing but one standard RCCL collective per step -- rank 0 writes the record as it stands to
        # STDERR (stdout stays the contract's ONE line at the end); a lease that dies inside a later captured-collective variant
        # still leaves the measured ones in the driver's stderr tail.
        if rank != 0 or lines.get(name) is None:
            return
        import json
        import sys
        try:
            best, summary = summarise(lines, errors)
        except AssertionError:
            return
        line = copy.deepcopy(lines[best])
        line['config']['exchange'] = best
        line['exchange_variants'] = {k: v for k, v in summary.items() if 'value' in v or k in errors}
        line['provisional'] = f'after variant {name!r}: {sum(1 for v in lines.values() if v is not None)} of {len(EXCHANGE_VARIANTS)} variants timed'
        print('[exchange auto] provisional ' + json.dumps(line), file=sys.stderr, flush=True)

    def bail(name):
        # a later variant hangs (a collective that never completes on this fabric) or dies: the record must not die with it.
        # Every rank's own timer ends its process; rank 0 first prints what the finished variants measured.
        try:
            emit(aborted=name)
        finally:
            os._exit(0)

    t_first = None
    t_auto0 = time.perf_counter()
    for name, flags in EXCHANGE_VARIANTS.items():
        # a wall-clock budget for the whole ranking: the record is ONE line at the very end, so a caller's time limit that fell
        # in the middle of a late variant would cost all of it.  Every rank takes the same decision (MAX of the clocks).
        if any(n_ not in errors for n_ in lines):  # (the same on every rank: errors are agreed on below; lines are rank 0's)
            el = torch.tensor([time.perf_counter() - t_auto0], dtype=torch.float64, device=dev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            if float(el.item()) > args.auto_budget:
                errors[name] = f'not run: the ranking had used {float(el.item()):.0f} s of its {args.auto_budget:.0f} s budget'
                lines[name] = None
                continue
        a = copy.copy(args)
        for k_, v_ in flags.items():
            setattr(a, k_, v_)
        a.ms_per_render, a.no_cpu_baseline = False, True
        have_one = any(v is not None for v in lines.values())
        timer = None
        if have_one:  # (the first variant runs unguarded: without it there is no record)
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
        provisional(name)
    emit()
