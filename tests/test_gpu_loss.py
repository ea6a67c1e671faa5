"""Fused L1 + SSIM loss kernel vs the plain-torch restatement of the reference graph (fp32 reference of the op).
Tolerance: loss value 1e-5 relative; gradient 1e-4 max-norm relative (separable vs 2-D window summation order)."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,W', [(64, 64), (100, 77), (800, 800), (16, 16), (7, 300)])
def test_fused_image_loss_matches_torch(H, W):
    from sk_gs_amd.losses import image_loss, image_loss_torch
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    y = torch.rand(3, H, W, generator=g).cuda()
    ref = image_loss_torch(x, y)
    (gref,) = torch.autograd.grad(ref * 3.0, x)
    out = image_loss(x, y)
    (got,) = torch.autograd.grad(out * 3.0, x)
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    assert rel_err(got, gref) <= 1e-4


def test_fused_image_loss_matches_reference_golden():
    """fused kernel vs values/gradients produced by the reference's own SSIM_Loss + ImageLoss (tests/golden/ssim.npz)"""
    import os
    import numpy as np
    from sk_gs_amd.losses import image_loss
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ssim.npz'))
    for k in range(3):
        x = torch.from_numpy(g[f'x{k}'])[0].permute(2, 0, 1).contiguous().cuda().requires_grad_(True)
        y = torch.from_numpy(g[f'y{k}'])[0].permute(2, 0, 1).contiguous().cuda()
        out = image_loss(x, y)
        (grad,) = torch.autograd.grad(out, x)
        assert abs(float(out.detach()) - float(g[f'total{k}'])) <= 1e-5 * float(g[f'total{k}'])
        assert rel_err(grad.permute(1, 2, 0), g[f'grad{k}'][0]) <= 1e-4


def test_fused_image_loss_identical_images():
    from sk_gs_amd.losses import image_loss
    x = torch.rand(3, 48, 48).cuda().requires_grad_(True)
    out = image_loss(x, x.detach().clone())
    assert abs(float(out)) < 1e-6
    out.backward()
    assert float(x.grad.abs().max()) < 1e-6
