#!/usr/bin/env python3
"""Where does the host time of one operator-path TRAINING step go (bench.py --autograd: model.render + image_loss +
backward + FusedAdam, config #1)?  cProfile over 200 iterations."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import _C, scene
from sk_gs_amd.losses import image_loss
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.optim import FusedAdam
from sk_gs_amd.renderer.gaussian_render import render

dev = torch.device('cuda')
if os.environ.get('SINGLE_THREAD_BACKWARD'):
    torch.autograd.set_multithreading_enabled(False)
P, W, H, V = 100_000, 800, 800, 4
model = SkinnedGaussians(P, 20, 5, sh_degree=3, num_frames=V, seed=0, deform_net=True, learn_joints=True).to(dev)  # bench.py's model
rs = [scene.raster_settings_from_camera(scene.make_camera(W, H, seed=v), sh_degree=3, colmap=True, device=dev) for v in range(V)]
targets = [torch.rand(3, H, W, device=dev) for _ in range(V)]
background = torch.ones(3, device=dev)
with torch.no_grad():
    _C.config.sync_num_rendered = True
    R = longest = 0
    for v in range(V):
        buf = render(**{k: t.detach() for k, t in model(v).items()}, raster_settings=rs[v])['buffer']
        R, longest = max(R, buf.R), max(longest, _C.read_status(buf.geomBuffer)['max_tile_count'])
_C.config.sync_num_rendered = False
_C.update_capacity_hint(P, W, H, int(R * 1.25), longest)
opt = FusedAdam(model.param_groups(lr=float(os.environ.get('LR', '1e-4'))), eps=1e-15, betas=(0.9, 0.999))


def one(i):
    v = i % V
    opt.zero_grad(set_to_none=False) if os.environ.get('ZERO') else None
    out = model.render(rs[v], time_id=v, background=background)
    loss = image_loss(out['images'], targets[v])
    loss.backward()
    opt.step()


for i in range(30):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue time {1e3 * (t1 - t0) / 200:.3f} ms / step, drained after {1e3 * (t2 - t1):.2f} ms more')
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    one(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(45)

# the backward runs on the autograd engine's thread: cProfile does not see it.  Wall-clock per wrapped entry point:
import collections, functools
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def timed(mod, name):
    f = getattr(mod, name)

    @functools.wraps(f)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t
            cnt[name] += 1
    setattr(mod, name, w)


import sk_gs_amd.deform as deform_mod
import sk_gs_amd.losses as losses_mod
import sk_gs_amd.renderer.gaussian_render as gr_mod
for cls in [v for v in vars(deform_mod).values() if isinstance(v, type) and issubclass(v, torch.autograd.Function)] + \
           [losses_mod._FusedImageLoss, gr_mod._RasterizeGaussians]:
    for m in ('forward', 'backward'):
        f = getattr(cls, m)
        def mk(f, label):
            def w(*a, **k):
                t = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    acc[label] += time.perf_counter() - t
                    cnt[label] += 1
            return staticmethod(w)
        setattr(cls, m, mk(f, f'{cls.__name__}.{m}'))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'with timers: host {1e3 * (t1 - t0) / 200:.3f} ms / step')
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f'  {k:45s} {1e6 * acc[k] / 200:8.1f} us / step  ({cnt[k] // 200} calls)')

if os.environ.get('TORCH_PROFILE'):
    # which torch operators (and so which launches) does one step contain, with gradients handed over (p.grad = None)?
    from torch.profiler import profile, ProfilerActivity
    params = [p for p in model.parameters() if p.requires_grad]

    def one_none(i):
        for p in params:
            p.grad = None
        v = i % V
        out = model.render(rs[v], time_id=v, background=background)
        image_loss(out['images'], targets[v]).backward()
        opt.step()

    for i in range(5):
        one_none(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(10):
            one_none(i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=70))
