#!/usr/bin/env python3
"""Host time of one forward-only operator-path render (the reference's FPS protocol, test.py:102-123) at config #1."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import _C, scene
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.renderer.gaussian_render import render

dev = torch.device('cuda')
P, W, H, V = 100_000, 800, 800, 4
model = SkinnedGaussians(P, 20, 5, sh_degree=3, num_frames=V, seed=0, deform_net=True, learn_joints=True).to(dev)
rs = [scene.raster_settings_from_camera(scene.make_camera(W, H, seed=v), sh_degree=3, colmap=True, device=dev) for v in range(V)]
bg = torch.ones(3, device=dev)
with torch.no_grad():
    R = longest = 0
    for v in range(V):
        buf = model.render(rs[v], time_id=v, background=bg)['buffer']
        R, longest = max(R, buf.R), max(longest, _C.read_status(buf.geomBuffer)['max_tile_count'])
    _C.config.sync_num_rendered = False
    _C.update_capacity_hint(P, W, H, int(R * 1.25), longest)
    for i in range(30):
        model.render(rs[i % V], time_id=i % V, background=bg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        model.render(rs[i % V], time_id=i % V, background=bg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host issue {1e3 * (t1 - t0) / 300:.3f} ms / render, drained after {1e3 * (t2 - t1):.2f} ms more -> {300 / (t2 - t0):.0f} renders/s')
    pr = cProfile.Profile()
    pr.enable()
    for i in range(300):
        model.render(rs[i % V], time_id=i % V, background=bg)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)
