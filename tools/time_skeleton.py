#!/usr/bin/env python3
"""The skeleton stage's two launches (network + kinematic chain) alone, graph-replayed: with / without the chain riding on
them, and the backward's phase stamps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd.deform_net import BoneChainDesc, FusedDeformMLP
from sk_gs_amd.model import SkinnedGaussians

torch.manual_seed(0)
M = 20
model = SkinnedGaussians(500, M, 4, sh_degree=0, num_frames=3, seed=5, deform_net=True, learn_joints=True).cuda()
mlp, topo = model.sk_deform_net, model.topology()
joints, t = model.joints.detach().contiguous(), torch.tensor([0.37], device='cuda')
gT = model.global_tr.detach()[1].contiguous()
f32 = dict(dtype=torch.float32, device='cuda')
net = mlp.dynamic_net
params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
heads = [torch.empty((M, 4), **f32), torch.empty((M, 4), **f32), torch.empty((M, 3), **f32)]
gh = [torch.randn((M, 4), **f32), torch.randn((M, 4), **f32), torch.randn((M, 3), **f32)]
grads, gx = [torch.zeros_like(p) for p in params], torch.zeros(M, net.in_channels, **f32)
b = BoneChainDesc()
b.M, b.root, b.num_levels = M, topo['root'], topo['num_levels']
b.parents, b.level_nodes, b.level_start = topo['parents'].data_ptr(), topo['level_nodes'].data_ptr(), topo['level_start'].data_ptr()
bone_T, chain_A, g_bone_T = torch.zeros(M, 7, **f32), torch.zeros(M, 7, **f32), torch.randn(M, 7, **f32)
g_j, g_g = torch.zeros(M, 3, **f32), torch.zeros(7, **f32)
b.joints, b.global_T, b.bone_T, b.chain_A = joints.data_ptr(), gT.data_ptr(), bone_T.data_ptr(), chain_A.data_ptr()
b.sk_r_raw, b.g_bone_T, b.g_joints, b.g_global_T = heads[0].data_ptr(), g_bone_T.data_ptr(), g_j.data_ptr(), g_g.data_ptr()
run = FusedDeformMLP(mlp, M)
for bones in (None, b):
    for what in ('fwd', 'bwd'):
        def body():
            if what == 'fwd':
                run.forward(joints, t, head_out=heads, bones=bones)
            else:
                run.backward(joints, t, gh, grads, gx, bones=bones)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run.forward(joints, t, head_out=heads, bones=bones)
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
        print(f'{"with" if bones is not None else "without"} chain, {what}: {us:.1f} us  {run.status()}')
run.workspace[8:12].view(torch.int32).fill_(1)
for _ in range(3):
    run.backward(joints, t, gh, grads, gx, bones=b)
torch.cuda.synchronize()
st = run.workspace[64:256].view(torch.int32).cpu().tolist()
for i, name in zip(range(13, 16), ('chain (8 hops)', 'input gradient', 'weight gradients')):
    print(f'backward {name:>18}: +{((st[2 * i] - st[2 * i - 2]) & 0xffffffff) * 10} ns')

# ---- the backward launch with the per-Gaussian rows' Adam update (221 MB) as its side job: launch time and the
# network's phases while the stream runs beside it
from sk_gs_amd.optim import FusedAdam
rows = [torch.nn.Parameter(torch.randn(100_000, n, **f32)) for n in (3, 3, 45, 1, 3, 4, 20)]
names = ['xyz', 'f_dc', 'f_rest', 'opacity', 'scaling', 'rotation', 'sp_W']
opt = FusedAdam([{'params': [p], 'lr': 1e-5, 'name': n} for p, n in zip(rows, names)])
side = opt.side_range(names)
run.workspace[8:12].view(torch.int32).fill_(0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run.backward(joints, t, gh, grads, gx, bones=b, side_adam=side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        for _ in range(20):
            run.backward(joints, t, gh, grads, gx, bones=b, side_adam=side)
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    print(f'with chain + side job (221 MB Adam), bwd: {(time.perf_counter() - t0) / 400 * 1e6:.1f} us  {run.status()}')
run.workspace[8:12].view(torch.int32).fill_(1)
for _ in range(3):
    run.backward(joints, t, gh, grads, gx, bones=b, side_adam=side)
torch.cuda.synchronize()
st = run.workspace[64:256].view(torch.int32).cpu().tolist()
for i, name in zip(range(13, 16), ('chain (8 hops)', 'input gradient', 'weight gradients')):
    print(f'  with side job, backward {name:>18}: +{((st[2 * i] - st[2 * i - 2]) & 0xffffffff) * 10} ns')
prev = 18   # the backward's prologue: stamps 18, 19, 21, 20 = entry, chain inputs staged, chain levels walked, weight slabs in LDS; 12 = prologue done
for i, name in ((22, 'every load back'), (19, 'chain inputs staged'), (21, 'chain levels walked'), (20, 'weight slabs in LDS'), (12, 'prologue done')):
    print(f'  with side job, backward prologue {name:>22}: +{((st[2 * i] - st[2 * prev]) & 0xffffffff) * 10} ns')
    prev = i

# ---- the rows' update split between the backward launch and the (next view's) forward launch: time of the pair
def pair_time(frac, P):
    rows = [torch.nn.Parameter(torch.randn(P, n, **f32)) for n in (3, 3, 45, 1, 3, 4, 20)]
    for p in rows:
        p.grad = torch.randn_like(p)
    opt = FusedAdam([{'params': [p], 'lr': 1e-5, 'name': n} for p, n in zip(rows, names)])
    head = opt.side_range(names, (0.0, frac)) if frac > 0 else None
    tail = opt.side_range(names, (frac, 1.0), after_advance=True) if frac < 1 else None
    def pair():
        run.backward(joints, t, gh, grads, gx, bones=b, side_adam=head)
        run.forward(joints, t, head_out=heads, bones=b, side_adam=tail)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pair()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(20):
                pair()
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 400 * 1e6
run.workspace[8:12].view(torch.int32).fill_(0)
for P in (100_000, 200_000, 500_000):
    print(f'P={P}: backward + forward launch pair, rows split:',
          ', '.join(f'{f:.2f}: {pair_time(f, P):.1f} us' for f in (1.0, 0.7, 0.6, 0.5)), run.status())
