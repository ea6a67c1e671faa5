"""host and GPU time per step of `bench.py --reference-loop <mode>` in blocks of 20 steps (no synchronisation inside the run)
usage (GPU box): python tools/step_series_reference_loop.py fused|accelerated|hooks [small]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from benchlib import options, reference_loop
mode = sys.argv[1]
small = len(sys.argv) > 2 and sys.argv[2] == 'small'     # (a scene whose GPU time is negligible: what is left is the host's time per step)
args = options.build_parser().parse_args(['--reference-loop', mode] + (['--config', '9'] if small else []))
s = reference_loop.setup(args, {**options.CONFIGS, 9: dict(name='small-4k-160', P=4000, M=20, K=5, W=160, H=120)})
N, B = 400, 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // B + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for b in range(N // B):
    t0 = time.perf_counter()
    for i in range(B):
        s.step(b * B + i)
    host.append((time.perf_counter() - t0) / B * 1e3)
    ev[b + 1].record()
torch.cuda.synchronize()
gpu = [ev[b].elapsed_time(ev[b + 1]) / B for b in range(N // B)]
print('host ms/step per block:', ' '.join(f'{h:.2f}' for h in host))
print('gpu  ms/step per block:', ' '.join(f'{g:.2f}' for g in gpu))
if s.rf is not None:
    r = s.rf.route_of_model(s.model)
    print('bucket', r._bucket, s.rf.calls)
