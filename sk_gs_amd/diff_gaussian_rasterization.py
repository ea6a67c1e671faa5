"""Variant (A) of the boundary: a drop-in for the upstream ``diff_gaussian_rasterization`` package, which is what the
reference's shipped configs actually render through (``use_official_gaussians_render=True``,
networks/gaussian_splatting.py:108,126-132,271-284; adapter networks/renderer/gaussian_render_origin.py:36-58).

Same constructor fields (including ``bg``), same call signature, rotations in **wxyz** order, returns
``(color[3,H,W] incl. background, radii)``.  The arithmetic is the in-tree ``colmap=True`` mode (a line-for-line
derivative of upstream) plus upstream's background term ``C + T * bg`` with ``T = 1 - opacity``; autograd
through that composition reproduces upstream's ``dL_dalpha += (-T_final / (1 - alpha)) * dot(bg, dL_dpixel)``.

To make ``from diff_gaussian_rasterization import ...`` resolve to this module: ``sk_gs_amd.install_as_diff_gaussian_rasterization()``.
"""
import sys
from typing import NamedTuple

import torch
from torch import nn

from sk_gs_amd.renderer import gaussian_render as _gr


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        s = self.raster_settings
        with torch.no_grad():
            return _gr.get_C_function('mark_visible')(positions, s.viewmatrix, s.projmatrix, True)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        s = self.raster_settings
        inner = _gr.GaussianRasterizationSettings(
            image_height=int(s.image_height), image_width=int(s.image_width), tanfovx=s.tanfovx, tanfovy=s.tanfovy,
            scale_modifier=s.scale_modifier, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, sh_degree=s.sh_degree,
            campos=s.campos, prefiltered=s.prefiltered, debug=s.debug, detach_other_extra=False, colmap=True)
        if rotations is not None:
            # (w, x, y, z) -> (x, y, z, w) as two slices: the backward of an index list is torch's sort-based index_put (0.3 ms for 100k rows)
            rotations = torch.cat([rotations[..., 1:], rotations[..., :1]], dim=-1)
        color, opacity, _, radii, _, _ = _gr.GaussianRasterizer(inner)(
            means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
            rotations=rotations, cov3D_precomp=cov3D_precomp)
        bg = s.bg.to(color.device, color.dtype).view(-1, 1, 1)
        return color + (1.0 - opacity)[None] * bg, radii


def install():
    """register this module as ``diff_gaussian_rasterization`` in ``sys.modules``"""
    sys.modules.setdefault('diff_gaussian_rasterization', sys.modules[__name__])
