"""Per-kernel HIP-event times of rasterize forward+backward on the static config-#1-sized scene.
usage: [SKGS_HIP_LIB=path] python tools/time_blend.py [P] [W] [H] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import torch
from helpers import scene_inputs
from test_gpu_raster import hip_forward, hip_backward
from sk_gs_amd import _C

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
act, rs, cam = scene_inputs(P, W, H, seed=0, device='cuda')
g = torch.Generator().manual_seed(1)
gc, go = torch.randn(3, H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
for it in range(3):
    fwd = hip_forward(act, rs)
    hip_backward(fwd, act, rs, gc, go)
torch.cuda.synchronize()
_C.profile_enable(None)
for it in range(reps):
    fwd = hip_forward(act, rs)
    hip_backward(fwd, act, rs, gc, go)
torch.cuda.synchronize()
res = _C.profile_collect()
print(os.environ.get('SKGS_HIP_LIB', 'default'), 'R =', fwd[0], ' '.join(f'{k}={ms / n * 1e3:.1f}us' for k, (ms, n) in res.items()))
