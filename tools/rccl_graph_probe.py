"""Can RCCL collectives be captured into a hipGraph on this stack?  A 1-rank group (one GPU): all_reduce and the in-place
all_gather_into_tensor are captured between two kernels and replayed.  usage (GPU box): timeout 120 python tools/rccl_graph_probe.py"""
import os, sys, time
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29561')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(x); torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y = x * 2
            dist.all_reduce(y)
            z = y + 1
        for _ in range(3):
            g.replay()
        s.synchronize()
    print('captured all_reduce in a graph: OK', float(z[0]))
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
# all_gather_into_tensor in place
try:
    buf = torch.zeros(4, 1024, device='cuda')
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            dist.all_gather_into_tensor(buf.view(-1)[:1024 * 1], buf[0].view(-1))
        g2.replay(); s.synchronize()
    print('captured all_gather_into_tensor: OK')
except Exception as e:
    print('capture 2 failed:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
