#!/usr/bin/env python3
"""Cost of one densification (clone + split + prune with their Adam-state surgery) at BASELINE config sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sk_gs_amd import densify
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.optim import FusedAdam

dev = torch.device('cuda')
for P, M in ((100_000, 20), (300_000, 20)):
    td, tp = [], []
    for rep in range(5):
        model = SkinnedGaussians(P, M, 5, sh_degree=3, num_frames=2, seed=0, deform_net=True).to(dev)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        opt = FusedAdam(model.param_groups(lr=1e-3), eps=1e-15)
        stats = densify.DensifyStats(P, dev)
        g = torch.Generator(device=dev).manual_seed(1)
        stats.xyz_gradient_accum = torch.rand(P, 1, device=dev, generator=g) * 4e-4   # ~half above max_grad 2e-4
        stats.denom = torch.ones(P, 1, device=dev)
        stats.max_radii2D = torch.rand(P, device=dev, generator=g) * 30
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        densify.densify(model, opt, stats, max_grad=2e-4, extent=4.0, generator=g)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        densify.prune(model, opt, stats, min_opacity=0.005, extent=4.0, max_screen_size=20.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        td.append(t1 - t0), tp.append(t2 - t1)
    print(f'P={P}: densify (clone + split) {1e3 * sorted(td)[2]:.2f} ms, prune {1e3 * sorted(tp)[2]:.2f} ms (medians of 5) -> {model.P} Gaussians')
