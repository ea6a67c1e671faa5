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


@pytest.mark.parametrize('seed', list(range(24)))
def test_fused_image_loss_odd_sizes(seed):
    """sizes around every boundary of the kernels' tiling: one pixel, narrower than the window, rows that are / are not
    16-byte aligned (the interior tiles' vector loads), one tile + a sliver, channel counts other than 3"""
    import numpy as np
    from sk_gs_amd.losses import image_loss, image_loss_torch
    r = np.random.RandomState(300 + seed)
    H = int(r.choice([1, 2, 5, 11, 31, 32, 33, 43, 64, 65, 97, 130]))
    W = int(r.choice([1, 3, 4, 8, 12, 31, 32, 36, 40, 44, 45, 48, 72, 73, 76, 132]))
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    y = torch.rand(3, H, W, generator=g).cuda()
    ref = image_loss_torch(x, y)
    (gref,) = torch.autograd.grad(ref, x)
    out = image_loss(x, y)
    (got,) = torch.autograd.grad(out, x)
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref)), (H, W)
    assert rel_err(got, gref) <= 1e-4, (H, W)


def test_fused_image_loss_unaligned_planes_take_the_scalar_path():
    """a prediction whose planes do not start on 16 bytes (a slice of a larger tensor): same values as the aligned call"""
    from sk_gs_amd.losses import image_loss
    g = torch.Generator().manual_seed(9)
    H, W = 96, 128
    big = torch.rand(3 * H * W + 3, generator=g).cuda()
    y = torch.rand(3, H, W, generator=g).cuda()
    x_un = big[1:1 + 3 * H * W].view(3, H, W)  # 4 bytes off
    assert x_un.data_ptr() % 16 != 0 and x_un.is_contiguous()
    x_al = x_un.clone().requires_grad_(True)
    x_un = x_un.detach().requires_grad_(True)
    a, b = image_loss(x_al, y), image_loss(x_un, y)
    (ga,), (gb,) = torch.autograd.grad(a, x_al), torch.autograd.grad(b, x_un)
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))
    assert rel_err(gb, ga) <= 1e-6
