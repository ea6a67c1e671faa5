"""Synthetic Gaussians, cameras and bones for tests and benchmarks (recipe: SURVEY.md section 8d / BASELINE.md).

Everything is generated on the CPU from ``torch.Generator().manual_seed(seed)`` so the same seed gives the same
scene on any box.  Camera helpers restate the reference's OpenCV conventions:
``look_at``      my_ext/ops_3d/coord_trans_opencv.py:87-119
``perspective``  my_ext/ops_3d/coord_trans_opencv.py:203-239
``fovx_to_fovy`` my_ext/ops_3d/coord_trans_common.py:56-60
``prepare_inputs`` (viewmatrix = Tw2v^T, projmatrix = (Tv2c @ Tw2v)^T, tanfov = tan(FoV/2))
                 networks/gaussian_splatting.py:247-300
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import Tensor
import torch.nn.functional as F

from sk_gs_amd.renderer.gaussian_render import GaussianRasterizationSettings


def look_at(eye: Tensor, at: Optional[Tensor] = None) -> Tensor:
    """world -> view (OpenCV: +z forward, +y down-ish as produced by the reference helper)"""
    eye = eye.float()
    at = torch.zeros_like(eye) if at is None else at.float()
    dir_vec = -F.normalize(eye - at, dim=-1)
    y_axis = dir_vec.new_tensor([0., -1., 0.])
    parallel = torch.linalg.cross(dir_vec, y_axis).norm() < 1e-6
    up = dir_vec.new_tensor([0., 0., -1.]) if bool(parallel) else dir_vec.new_tensor([0., 1., 0.])
    right_vec = -F.normalize(torch.linalg.cross(up, dir_vec), dim=-1)
    up_vec = torch.linalg.cross(dir_vec, right_vec)
    R = torch.eye(4)
    T = torch.eye(4)
    R[0, :3], R[1, :3], R[2, :3] = right_vec, up_vec, dir_vec
    T[:3, 3] = -eye
    return R @ T


def fovx_to_fovy(fovx: float, aspect: float = 1.) -> float:
    return math.atan(math.tan(fovx * 0.5) / aspect) * 2.0


def perspective(fovy: float, n: float, f: float, size) -> Tensor:
    """OpenCV-style view -> clip matrix (w = +z)"""
    aspect = size[0] / size[1]
    y = math.tan(fovy * 0.5)
    x = y * aspect
    top, right = y * n, x * n
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * n / (right - left)
    P[1, 1] = 2.0 * n / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = (f + n) / (f - n)
    P[2, 3] = -(2 * f * n) / (f - n)
    return P


def make_camera(width: int, height: int, seed: int = 0, radius: float = 4.0, fovx: float = 0.6911, near: float = 2.,
                far: float = 6., eye: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """A camera on a sphere looking at the origin; returns the reference's ``infos`` dict fields
    (datasets/DNerfDataset.py:231-261): Tw2v, Tv2c, FoV, campos, size."""
    g = torch.Generator().manual_seed(1000 + seed)
    if eye is None:
        d = F.normalize(torch.randn(3, generator=g), dim=0)
        d[1] = d[1] * 0.5  # keep away from the poles
        eye = F.normalize(d, dim=0) * radius
    aspect = width / height
    fovy = fovx_to_fovy(fovx, aspect)
    Tw2v = look_at(eye)
    Tv2c = perspective(fovy, near, far, (width, height))
    return dict(Tw2v=Tw2v, Tv2c=Tv2c, FoV=torch.tensor([fovx, fovy]), campos=eye.clone(), size=(width, height))


def raster_settings_from_camera(cam: Dict[str, Tensor], sh_degree: int = 3, colmap: bool = True,
                                scale_modifier: float = 1.0, device=None, debug=False) -> GaussianRasterizationSettings:
    """What ``GaussianSplatting.prepare_inputs`` builds (gaussian_splatting.py:271-284) for ``colmap=True``; for
    ``colmap=False`` the row-major matrices the in-tree kernels expect (gui.py:552-567)."""
    Tw2v, Tv2c = cam['Tw2v'], cam['Tv2c']
    full = Tv2c @ Tw2v
    W, H = cam['size']
    if colmap:
        view, proj = Tw2v.transpose(-1, -2).contiguous(), full.transpose(-1, -2).contiguous()
    else:
        view, proj = Tw2v.contiguous(), full.contiguous()
    dev = device if device is not None else Tw2v.device
    return GaussianRasterizationSettings(
        image_height=H, image_width=W,
        tanfovx=math.tan(0.5 * float(cam['FoV'][0])), tanfovy=math.tan(0.5 * float(cam['FoV'][1])),
        scale_modifier=scale_modifier, viewmatrix=view.to(dev), projmatrix=proj.to(dev), sh_degree=sh_degree,
        campos=cam['campos'].to(dev), prefiltered=False, debug=debug, colmap=colmap)


def make_gaussians(P: int, seed: int = 0, sh_degree: int = 3, scale_mult: float = 1.0) -> Dict[str, Tensor]:
    """Raw (pre-activation) Gaussian parameters, the tensors GaussianSplatting keeps
    (networks/gaussian_splatting.py:134-139)."""
    g = torch.Generator().manual_seed(seed)
    S = (sh_degree + 1) ** 2
    sigma0 = 0.01 * (1e5 / max(P, 1)) ** (1. / 3.) * scale_mult
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.3
    log_scale = math.log(sigma0) + 0.3 * torch.randn(P, 3, generator=g)
    rot = F.normalize(torch.randn(P, 4, generator=g), dim=-1)
    opacity_logit = 1.5 * torch.randn(P, 1, generator=g)
    sh = torch.cat([torch.randn(P, 1, 3, generator=g), 0.1 * torch.randn(P, S - 1, 3, generator=g)], dim=1)
    return dict(xyz=xyz, log_scale=log_scale, rot=rot, opacity_logit=opacity_logit, sh=sh.contiguous())


def activate(params: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """static-stage activations (sk_gs.py:1162,1192,1202-1203 with d_* = 0)"""
    return dict(means3D=params['xyz'], scales=torch.exp(params['log_scale']),
                rotations=F.normalize(params['rot'], dim=-1), opacity=torch.sigmoid(params['opacity_logit']),
                sh=params['sh'])


def make_bones(M: int, seed: int = 0) -> Dict[str, Tensor]:
    """Joints in U(-1,1)^3, a random tree (parent(i) in [0,i)), small local rotations, tiny d_rot / d_scale."""
    g = torch.Generator().manual_seed(2000 + seed)
    joints = torch.rand(M, 3, generator=g) * 2 - 1
    parents = torch.zeros(M, dtype=torch.long)
    for i in range(1, M):
        parents[i] = int(torch.randint(0, i, (1,), generator=g))
    axis_angle = 0.2 * torch.randn(M, 3, generator=g)
    d_rot = 1e-2 * torch.randn(M, 4, generator=g)
    d_scale = 1e-3 * torch.randn(M, 3, generator=g)
    return dict(joints=joints, parents=parents, axis_angle=axis_angle, d_rot=d_rot, d_scale=d_scale)
