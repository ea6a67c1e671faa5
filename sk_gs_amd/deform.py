"""LBS deform operator (the "deform boundary" of SURVEY.md section 8b).

``lbs_deform`` replaces the op sequence of ``SkeletonGaussianSplatting.sk_stage`` + the activation epilogue of
``forward`` (networks/sk_gs.py:1143-1150,1162,1192-1203) -- ~14 torch/lietorch kernels with [P,K,7] / [P,K,3]
temporaries -- by ONE fused HIP kernel per direction (csrc/deform.hip).

Gradient convention for ``bone_T`` ([M,7] = t, q_xyzw): plain Euclidean gradients w.r.t. the 7 stored numbers,
including the Jacobian of the quaternion normalisation the SE3 constructor applies (lie.h:45-47).  For a unit q this
equals what lietorch's ``FromVec`` returns (tangent gradient times pinv of ``orthogonal_projector``, lie.h:82-90,
303-311): both are the unique gradient of a scale-invariant function of q, orthogonal to q.
"""
import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from sk_gs_amd import _C


class _LBSDeform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit):
        means, scales, rotations, opacity, _, _, _ = _C.lbs_deform_forward(
            points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit)
        ctx.save_for_backward(points, weights, indices, bone_T, bone_drot, bone_dscale, log_scale, rot, opacity_logit)
        ctx.mark_non_differentiable(indices)
        return means, scales, rotations, opacity

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_means, g_scales, g_rotations, g_opacity):
        points, weights, indices, bone_T, bone_drot, bone_dscale, log_scale, rot, opacity_logit = ctx.saved_tensors
        P = points.shape[0]
        dev = points.device

        def z(g, *shape):
            return torch.zeros(shape, device=dev) if g is None else g

        g_w, g_T, g_dr, g_ds, g_xyz, g_ls, g_rot, g_op = _C.lbs_deform_backward(
            points, weights, indices, bone_T, bone_drot, bone_dscale, log_scale, rot, opacity_logit,
            z(g_means, P, 3), z(g_scales, P, 3), z(g_rotations, P, 4), z(g_opacity, P, 1))
        # points is the detached copy of xyz (sk_gs.py:1113): no gradient
        return None, g_w, None, g_T, g_dr, g_ds, g_xyz, g_ls, g_rot, g_op


def lbs_deform(points: Tensor, weights: Tensor, indices: Tensor, bone_T: Tensor, bone_drot: Tensor,
               bone_dscale: Tensor, xyz: Tensor, log_scale: Tensor, rot: Tensor, opacity_logit: Tensor
               ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """``(means, scales, rotations, opacity)`` of the deformed, activated Gaussians.

    ``points`` must be ``xyz.detach()`` (the reference detaches before skinning); ``indices`` int64 [P,K]."""
    return _LBSDeform.apply(points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot,
                            opacity_logit)


class _KnnSoftmaxWeights(torch.autograd.Function):
    """``knn_points`` + ``softmax(gather(sp_W, idx))`` (the `W` method of calc_LBS_weight, sk_gs.py:757,767-768) as one
    launch per direction: ``skgs_knn_lbs_weights`` forward, ``skgs_lbs_weights_backward`` (dense [P,M] logit gradient)
    backward -- the kernels ``FusedViewStep`` uses, so both paths see the same weights, bit for bit."""

    @staticmethod
    def forward(ctx, points, joints, sp_W, K: int):
        lib = _C.load_library()
        _C._require_gpu(points, 'points')
        dev = points.device
        with _C._on_device(dev):
            pts, jts, logits = _C._f32c(points, dev), _C._f32c(joints, dev), _C._f32c(sp_W, dev)
            P, M = logits.shape
            idx = torch.empty((P, K), dtype=torch.int64, device=dev)
            w = torch.empty((P, K), dtype=torch.float32, device=dev)
            _C._check(lib.skgs_knn_lbs_weights(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_void_p(_C._ptr(pts)),
                                               C.c_void_p(_C._ptr(jts)), C.c_void_p(_C._ptr(logits)),
                                               C.c_void_p(_C._ptr(idx)), C.c_void_p(_C._ptr(w)), _C._stream()))
        ctx.save_for_backward(w, idx)
        ctx.M = M
        ctx.mark_non_differentiable(idx)
        return w, idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, _g_idx):
        w, idx = ctx.saved_tensors
        lib = _C.load_library()
        P, K = w.shape
        with _C._on_device(w.device):
            g_w = _C._f32c(g_w, w.device)
            g_sp_W = torch.empty((P, ctx.M), dtype=torch.float32, device=w.device)
            _C._check(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(ctx.M), C.c_int32(K), C.c_void_p(_C._ptr(w)),
                                                    C.c_void_p(_C._ptr(idx)), C.c_void_p(_C._ptr(g_w)),
                                                    C.c_void_p(_C._ptr(g_sp_W)), _C._stream()))
        return None, None, g_sp_W, None


def calc_lbs_weight(points: Tensor, joints: Tensor, K: int, sp_W: Optional[Tensor] = None,
                    kernel_radius: Optional[Tensor] = None, kernel_weight: Optional[Tensor] = None,
                    temperature: float = 1., feature: Optional[Tensor] = None, sp_feature: Optional[Tensor] = None
                    ) -> Tuple[Tensor, Tensor]:
    """``calc_LBS_weight`` of the reference (networks/sk_gs.py:751-774): K nearest bones + one of three weightings.
    The P x M nearest-neighbour search runs in the HIP ``knn_bones`` kernel (pytorch3d.knn_points semantics)."""
    if feature is not None and sp_feature is not None:
        points = torch.cat([points.detach(), feature], dim=-1)
        joints_q = torch.cat([joints.detach(), sp_feature], dim=-1)
    else:
        joints_q = joints
        # plain `W` method on xyz: search + gather + softmax in one launch (joint table in LDS: a few thousand bones at most)
        if sp_W is not None and kernel_radius is None and points.shape[-1] == 3 and K <= 16 and K <= sp_W.shape[1] <= 2048:
            return _KnnSoftmaxWeights.apply(points.detach(), joints.detach(), sp_W, K)
    with torch.no_grad():
        _, indices = _C.knn_bones(points.detach(), joints_q.detach(), K)
    if kernel_radius is not None or (sp_W is None):
        # distances must carry gradients to the joints for these two methods: recompute them in torch on [P,K]
        nn_dist = (points[:, None, :] - joints_q[indices]).square().sum(-1)
    if kernel_radius is not None:
        radius = kernel_radius[indices]
        weights = torch.exp(-nn_dist / (2 * radius ** 2))
        if kernel_weight is not None:
            weights = weights * kernel_weight[indices]
        weights = weights + 1e-7
        weights = weights / weights.sum(dim=-1, keepdim=True)
    elif sp_W is not None:
        weights = torch.gather(sp_W, dim=1, index=indices).softmax(dim=-1)
    else:
        weights = torch.softmax(-nn_dist / temperature, dim=-1)
    return weights, indices
