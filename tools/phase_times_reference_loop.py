"""wall time of each phase of one iteration of `bench.py --reference-loop fused` on the HOST (a tiny scene: the GPU never is the limit)
usage (GPU box): python tools/phase_times_reference_loop.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchlib import options, reference_loop
args = options.build_parser().parse_args(['--reference-loop', 'fused', '--config', '9'])
s = reference_loop.setup(args, {**options.CONFIGS, 9: dict(name='small-4k-160', P=4000, M=20, K=5, W=160, H=120)})
rf, opt = s.rf, s.opt
for i in range(150):
    s.step(i)
torch.cuda.synchronize()
names = ['render', 'loss', 'sum', 'backward', 'opt.step', 'zero_grad']
acc = [0.0] * len(names)
N = 300
for i in range(N):
    v = i % 3
    t = [time.perf_counter()]
    out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v]); t.append(time.perf_counter())
    losses = s.model_loss(out, s.targets_hwc[v]); t.append(time.perf_counter())
    loss = sum(losses.values()); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    for k in range(len(names)):
        acc[k] += t[k + 1] - t[k]
torch.cuda.synchronize()
print(' | '.join(f'{n} {1e6 * a / N:.0f} us' for n, a in zip(names, acc)), '| total', f'{1e6 * sum(acc) / N:.0f} us')
