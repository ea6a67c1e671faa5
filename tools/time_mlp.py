#!/usr/bin/env python3
"""Time the deform network: fused one-launch kernels (NCOL 4 / 8) vs the per-layer kernels, eager and graph-replayed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import _C
from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP, DeformMLPRunner

torch.manual_seed(0)
mlp = DeformMLP().cuda()
lib = _C.load_library()
for B in (20, 32):
    joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.3], device='cuda'), torch.randn(B, 11, device='cuda')
    params = [p for l in mlp.dynamic_net.net for p in (l.weight, l.bias)] + [mlp.dynamic_net.last_weight, mlp.dynamic_net.last_bias]
    grads = [torch.zeros_like(p) for p in params]
    for ncol in (8,):
        run = FusedDeformMLP(mlp, B)
        for what in ('fwd', 'bwd', 'both'):
            def body():
                if what in ('fwd', 'both'):
                    run.forward(joints, t)
                if what in ('bwd', 'both'):
                    run.backward(joints, t, g, grads)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                run.forward(joints, t)
                body()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):
                    for _ in range(20):
                        body()
                graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    graph.replay()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 400 * 1e6
            print(f'B={B} ncol={ncol} fused {what}: {us:.1f} us per pass (graph of 20), status {run.status()}')

# ---- in-kernel phase stamps of workgroup 0 (forward): {100 MHz real-time counter, shader clock}
B = 20
joints, t = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.3], device='cuda')
run = FusedDeformMLP(mlp, B)
run.forward(joints, t)
run.workspace[8:12].view(torch.int32).fill_(1)
for _ in range(3):
    run.forward(joints, t)
torch.cuda.synchronize()
st = run.workspace[64:256].view(torch.int32).cpu().tolist()
names = ['prologue', 'layer0'] + [n for l in range(1, 9) for n in (f'gather{l}', f'layer{l}')]
print(f'entry -> first stamp (prologue): {((st[0] - st[46]) & 0xffffffff) * 10} ns')
for i in range(1, len(names)):
    dt, dc = (st[2 * i] - st[2 * i - 2]) & 0xffffffff, (st[2 * i + 1] - st[2 * i - 1]) & 0xffffffff
    print(f'{names[i]:>10}: +{dt * 10} ns  {dc} clk  ({dc / max(dt, 1) * 100:.0f} MHz)')

# ---- backward: stamps 12..15 of workgroup 0 = prologue done, chain done, input gradient done, weight gradients done
g = torch.randn(B, 11, device='cuda')
params = [p for l in mlp.dynamic_net.net for p in (l.weight, l.bias)] + [mlp.dynamic_net.last_weight, mlp.dynamic_net.last_bias]
grads = [torch.zeros_like(p) for p in params]
g_x0 = torch.zeros(B, mlp.dynamic_net.in_channels, device='cuda')
for _ in range(3):
    run.backward(joints, t, g, grads, g_x0)
torch.cuda.synchronize()
st = run.workspace[64:256].view(torch.int32).cpu().tolist()
for i, name in zip(range(13, 16), ('chain (8 hops)', 'input gradient', 'weight gradients')):
    print(f'backward {name:>18}: +{((st[2 * i] - st[2 * i - 2]) & 0xffffffff) * 10} ns')
# the backward's prologue: stamps 18, 19, 21, 20 = entry, chain inputs staged, chain levels walked, weight slabs in LDS; 12 = prologue done
seq = [(19, 'chain inputs staged'), (21, 'chain levels walked'), (20, 'weight slabs in LDS'), (12, 'prologue done (gZ of the last layer, re-poison)')]
prev = 18
for i, name in seq:
    print(f'backward prologue {name:>48}: +{((st[2 * i] - st[2 * prev]) & 0xffffffff) * 10} ns')
    prev = i
