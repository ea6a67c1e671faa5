"""Fused multi-tensor Adam kernel vs torch.optim.Adam (same hyper-parameters as the reference: eps 1e-15).
Tolerance: 1e-6 max-norm relative on parameters and both moments after 6 steps."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch():
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.view_parallel import FlatGradBuffer
    gen = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (37,), (5, 7), (4099,)]  # incl. unaligned offsets
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 1e-3, 1e-3, 1e-3]
    a = [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam([{'params': [p], 'lr': lr} for p, lr in zip(a, lrs)], eps=1e-15, betas=(0.9, 0.999))
    fb = FlatGradBuffer(b)  # gradients as views of one flat buffer (some of them not 16-byte aligned)
    opt = FusedAdam([{'params': [p], 'lr': lr} for p, lr in zip(b, lrs)], eps=1e-15, betas=(0.9, 0.999))
    for step in range(6):
        for p, q in zip(a, b):
            g = torch.randn(p.shape, generator=gen).cuda() * (0.0 if step == 3 else 1.0)  # one all-zero gradient step
            p.grad = g.clone()
            q.grad.copy_(g)
        ref.step()
        opt.step()
    assert float(opt.step_count.item()) == 6.0
    for p, q in zip(a, b):
        assert rel_err(q, p) <= 1e-6
        assert rel_err(opt.state[q]['exp_avg'], ref.state[p]['exp_avg']) <= 1e-6
        assert rel_err(opt.state[q]['exp_avg_sq'], ref.state[p]['exp_avg_sq']) <= 1e-6
    opt.set_lr(0, 0.5)
    before = b[0].detach().clone()
    opt.step()
    assert float((b[0] - before).abs().max()) > 1e-3


def test_fused_adam_clears_the_requested_gradient_span_after_the_update():
    """zero_after_step: the launch that bumps the step counter also clears a gradient span (FusedViewStep's per-frame
    table gradients), after the update has consumed it"""
    from sk_gs_amd.optim import FusedAdam
    flat = torch.ones(10 + 37, device='cuda')
    a, b = torch.nn.Parameter(torch.zeros(10, device='cuda')), torch.nn.Parameter(torch.zeros(37, device='cuda'))
    a.grad, b.grad = flat[:10], flat[10:]
    opt = FusedAdam([{'params': [a, b], 'lr': 0.1}], zero_after_step=flat[10:])
    opt.step()
    torch.cuda.synchronize()
    assert float(a.grad.min()) == 1.0 and float(b.grad.abs().max()) == 0.0
    assert float(b.data.max()) < 0.0 and float(opt.step_count.item()) == 1.0  # b was updated with the gradient first
