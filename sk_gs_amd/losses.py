"""Image loss of the training step: ``0.8 * L1 + 0.2 * (1 - SSIM)`` (exps/default.yaml:83-84,
networks/sk_gs.py:1524-1529, networks/losses/image_loss.py:6-32, networks/losses/ssim.py:20-62).

``image_loss`` is the fused HIP kernel pair of scope row (f)-1 (csrc/image_loss.hip); ``image_loss_torch`` is the plain
torch restatement (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) kept as the fp32 numerics
reference of its tests.
"""
import math

import torch
import torch.nn.functional as F
from torch import Tensor

_window_cache = {}


def gaussian_window(window_size: int = 11, sigma: float = 1.5, channels: int = 3, device=None) -> Tensor:
    key = (window_size, sigma, channels, str(device))
    w = _window_cache.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        w = w2.expand(channels, 1, window_size, window_size).contiguous().to(device)
        _window_cache[key] = w
    return w


def ssim_map(img1: Tensor, img2: Tensor, window_size: int = 11) -> Tensor:
    """img*: [B, C, H, W]"""
    C = img1.shape[-3]
    window = gaussian_window(window_size, 1.5, C, img1.device).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=C)
    mu2 = F.conv2d(img2, window, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


def ssim_loss(img1_chw: Tensor, img2_chw: Tensor) -> Tensor:
    """1 - mean SSIM, images [C,H,W] or [B,C,H,W]"""
    if img1_chw.ndim == 3:
        img1_chw, img2_chw = img1_chw[None], img2_chw[None]
    return 1.0 - ssim_map(img1_chw, img2_chw).mean()


def image_loss_torch(pred_chw: Tensor, gt_chw: Tensor, lambda_l1: float = 0.8, lambda_ssim: float = 0.2) -> Tensor:
    """plain-torch restatement (reference graph): used as the fp32 numerics reference of the fused kernel"""
    l1 = (pred_chw - gt_chw).abs().mean()
    return lambda_l1 * l1 + lambda_ssim * ssim_loss(pred_chw, gt_chw)


class _FusedImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, lambda_l1, lambda_ssim):
        from sk_gs_amd import _C
        loss3, ws = _C.image_loss_forward(pred, gt, lambda_l1, lambda_ssim)
        ctx.save_for_backward(pred, gt, ws)
        ctx.lambdas = (lambda_l1, lambda_ssim)
        return loss3[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        from sk_gs_amd import _C
        pred, gt, ws = ctx.saved_tensors
        g = _C.image_loss_backward(pred, gt, ctx.lambdas[0], ctx.lambdas[1], grad_loss.reshape(1), ws)
        return g, None, None, None


def image_loss(pred_chw: Tensor, gt_chw: Tensor, lambda_l1: float = 0.8, lambda_ssim: float = 0.2) -> Tensor:
    """``lambda_l1 * L1 + lambda_ssim * (1 - SSIM)``: one fused HIP kernel per direction (csrc/image_loss.hip).
    GPU tensors only -- there is no fallback; ``image_loss_torch`` is the explicit torch reference used by tests."""
    return _FusedImageLoss.apply(pred_chw, gt_chw, float(lambda_l1), float(lambda_ssim))
