"""Host-side profile of `bench.py --reference-loop <mode>`: where the Python time of one iteration goes (cProfile, top entries by
cumulative time), and how much of the iteration is host work (time to ISSUE n steps without waiting) against the GPU's own time.
usage (GPU box): python tools/profile_reference_loop.py [fused|accelerated|hooks] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from benchlib import options, reference_loop  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fused'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
args = options.build_parser().parse_args(['--reference-loop', mode])
s = reference_loop.setup(args, options.CONFIGS)
for i in range(10):
    s.step(i)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(n):
        s.step(10 + i)
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'{mode}: {n} steps issued in {1e3 * t_issue / n:.3f} ms / step (host), finished in {1e3 * t_all / n:.3f} ms / step; '
          f'GPU events {e0.elapsed_time(e1) / n:.3f} ms / step')
if os.environ.get('SKGS_NO_CPROFILE'):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    s.step(10 + n + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(25)
