"""time the superpoint deform network's launches (csrc/sp_mlp.hip) with HIP events: forward, backward (rows + weights)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd.superpoint import SpDeformNet

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = 200
torch.manual_seed(0)
net = SpDeformNet().cuda()
x = (torch.rand(M, 3) * 2 - 1).cuda()
t = torch.tensor([0.3], device='cuda')
run = net.runner(M)
for p in net.parameters():
    p.grad = torch.zeros_like(p)
gT, gr, gs = torch.randn(M, 7).cuda(), torch.randn(M, 4).cuda(), torch.randn(M, 3).cuda()


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'M = {M}')
print(f'forward            {timed(lambda: run.forward(x, t)):8.1f} us')
print(f'backward (2 launches) {timed(lambda: run.backward(gT, gr, gs)):8.1f} us')
flops_f = 2 * M * (93 * 256 + 6 * 256 * 256 + 349 * 256 + 256 * 10)
print(f'forward GEMM flops {flops_f / 1e9:.3f} G')
