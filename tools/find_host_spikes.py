"""per-step host time of `bench.py --reference-loop fused` and the garbage collections behind its spikes
usage (GPU box): python tools/find_host_spikes.py"""
import os, sys, time, gc
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from benchlib import options, reference_loop
args = options.build_parser().parse_args(['--reference-loop', 'fused'])
s = reference_loop.setup(args, options.CONFIGS)
events = []
def cb(phase, info):
    events.append((time.perf_counter(), phase, info.get('generation'), info.get('collected')))
gc.callbacks.append(cb)
print("frozen objects", gc.get_freeze_count())
ts = []
for i in range(600):
    t0 = time.perf_counter()
    s.step(i)
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
big = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 2.0]
print('steps over 2 ms:', big)
t_start = None
for k in range(0, len(events) - 1, 2):
    (ta, pa, g, _), (tb, pb, _, col) = events[k], events[k + 1]
    if tb - ta > 0.002:
        print('gc gen', g, 'took', round((tb - ta) * 1e3, 1), 'ms, collected', col)
print('gc counts', gc.get_count(), gc.get_threshold(), len(gc.get_objects()))
