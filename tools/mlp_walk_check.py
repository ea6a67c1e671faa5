"""the fused deform network against torch over a bounded random walk of weights and inputs (eager launches, one workspace): prints the
worst relative error per iteration block -- SKGS_HIP_LIB=<older build> runs the same walk on another build of the kernels
usage (GPU box): python tools/mlp_walk_check.py [iterations=96]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import rel_err
from test_gpu_deform_net import _ref_with_input_grad
from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
torch.manual_seed(5)
mlp = DeformMLP().cuda()
B = 20
run = FusedDeformMLP(mlp, B)
for it in range(n):
    with torch.no_grad():
        for p in mlp.parameters():
            p.mul_(0.97).add_(torch.randn_like(p) * 0.02)
        mlp.dynamic_net.last_weight.normal_(0, 0.1)
    joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.rand(1, device='cuda'), torch.randn(B, 11, device='cuda')
    ref_out, ref_acts, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
    grads, g_x0 = [torch.zeros_like(r) for r in ref_grads], torch.zeros_like(ref_gx0)
    out = run.forward(joints, t)
    run.backward(joints, t, g, grads, g_x0)
    errs = [float(rel_err(out, ref_out)), float(rel_err(run.acts, ref_acts)), float(rel_err(g_x0, ref_gx0))] + [float(rel_err(a, r)) for a, r in zip(grads, ref_grads)]
    if max(errs) > 1e-4 or it % 16 == 0:
        dead = [int((ref_acts[l] <= 0).all(dim=0).sum()) for l in range(ref_acts.shape[0])]
        flips = [int(((run.acts[l] > 0) != (ref_acts[l] > 0)).sum()) for l in range(ref_acts.shape[0])]
        print('   ReLU decisions that differ between the kernel\'s forward and torch\'s, per layer:', flips)
        print(it, 'worst', f'{max(errs):.2e}', 'fwd', f'{errs[0]:.1e} {errs[1]:.1e}', 'g_x0', f'{errs[2]:.1e}', 'grads', ' '.join(f'{e:.0e}' for e in errs[3:]),
              '| units dead on every row, per layer:', dead, '| |w|', f'{float(mlp.dynamic_net.net[3].weight.abs().mean()):.3f}')
