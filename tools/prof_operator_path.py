#!/usr/bin/env python3
"""Where does the host time of one operator-path render + backward go?  cProfile over 300 iterations (config #1)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import _C, scene
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.renderer.gaussian_render import render

dev = torch.device('cuda')
P, W, H = 100_000, 800, 800
model = SkinnedGaussians(P, 20, 5, sh_degree=3, num_frames=2, seed=0).to(dev)
rs = scene.raster_settings_from_camera(scene.make_camera(W, H, seed=0), sh_degree=3, colmap=True, device=dev)
with torch.no_grad():
    net = {k: v.detach() for k, v in model(0).items()}
    _C.config.sync_num_rendered = True
    buf = render(**net, raster_settings=rs)['buffer']
    R, longest = buf.R, _C.read_status(buf.geomBuffer)['max_tile_count']
_C.config.sync_num_rendered = False
_C.update_capacity_hint(P, W, H, int(R * 1.25), 0 if os.environ.get('SKGS_COMPACT') else longest)
gcol, gop = torch.randn(3, H, W, device=dev), torch.randn(H, W, device=dev)
ins = {k: v.clone().requires_grad_(True) for k, v in net.items()}


def one():
    o = render(**ins, raster_settings=rs)
    torch.autograd.backward([o['images'], o['opacity']], [gcol, gop])


for _ in range(30):
    one()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    one()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue time {1e3 * (t1 - t0) / 300:.3f} ms / iteration, drained after {1e3 * (t2 - t1):.2f} ms more')
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
