import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import torch
from helpers import scene_inputs
from test_gpu_raster import hip_forward, hip_backward
from sk_gs_amd import _C
for (P, W, H) in [(100000, 800, 800), (500000, 1024, 1024), (200000, 512, 512), (300000, 800, 800)]:
    act, rs, cam = scene_inputs(P, W, H, seed=0, device='cuda')
    g = torch.Generator().manual_seed(1)
    gc, go = torch.randn(3, H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
    for ppl in (1, 2, 4):
        _C.set_pixels_per_lane(ppl)
        for it in range(3):
            fwd = hip_forward(act, rs); hip_backward(fwd, act, rs, gc, go)
        torch.cuda.synchronize()
        _C.profile_enable(None)
        for it in range(10):
            fwd = hip_forward(act, rs); hip_backward(fwd, act, rs, gc, go)
        torch.cuda.synchronize()
        res = _C.profile_collect(); _C.profile_enable([])
        print(P, W, H, 'ppl', ppl, 'R', fwd[0], ' '.join(f'{k}={ms / n * 1e3:.0f}' for k, (ms, n) in res.items() if 'render' in k), flush=True)
