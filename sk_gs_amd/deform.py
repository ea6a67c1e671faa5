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


class _KnnDistWeights(torch.autograd.Function):
    """``knn_points`` + the `weighted_kernel` / `kernel` / `dist` weighting of calc_LBS_weight (sk_gs.py:757-766,770) as ONE
    launch per direction (``skgs_knn_dist_weights_forward/backward``): search, weights, and in the backward the gradients
    autograd hands to the points / joints (through the distances), ``kernel_radius`` and ``kernel_weight``."""

    @staticmethod
    def forward(ctx, points, joints, kernel_radius, kernel_weight, temperature: float, K: int):
        lib = _C.load_library()
        _C._require_gpu(points, 'points')
        dev = points.device
        with _C._on_device(dev):
            pts, jts = _C._f32c(points, dev), _C._f32c(joints, dev)
            rad = None if kernel_radius is None else _C._f32c(kernel_radius, dev)
            kw = None if kernel_weight is None else _C._f32c(kernel_weight, dev)
            P, dim = pts.shape
            M = jts.shape[0]
            idx = torch.empty((P, K), dtype=torch.int64, device=dev)
            w = torch.empty((P, K), dtype=torch.float32, device=dev)
            dist = torch.empty((P, K), dtype=torch.float32, device=dev)
            _C._check(lib.skgs_knn_dist_weights_forward(
                C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(dim), C.c_void_p(_C._ptr(pts)), C.c_void_p(_C._ptr(jts)),
                C.c_void_p(_C._ptr(rad)), C.c_void_p(_C._ptr(kw)), C.c_float(float(temperature)), C.c_int32(0),
                C.c_void_p(_C._ptr(idx)), C.c_void_p(_C._ptr(w)), C.c_void_p(_C._ptr(dist)), _C._stream()))
        ctx.save_for_backward(pts, jts, rad, kw, w, idx, dist)
        ctx.temperature = float(temperature)
        ctx.mark_non_differentiable(idx)
        return w, idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, _g_idx):
        pts, jts, rad, kw, w, idx, dist = ctx.saved_tensors
        lib = _C.load_library()
        dev = w.device
        P, K = w.shape
        M, dim = jts.shape
        need_p, need_j, need_r, need_k = ctx.needs_input_grad[:4]
        with _C._on_device(dev):
            g_w = _C._f32c(g_w, dev)
            g_p = torch.empty((P, dim), dtype=torch.float32, device=dev) if need_p else None
            g_j = torch.empty((M, dim), dtype=torch.float32, device=dev) if need_j else None
            g_r = torch.empty((M,), dtype=torch.float32, device=dev) if (need_r and rad is not None) else None
            g_k = torch.empty((M,), dtype=torch.float32, device=dev) if (need_k and kw is not None) else None
            lib.skgs_knn_dist_weights_workspace_bytes.restype = C.c_size_t
            nbytes = int(lib.skgs_knn_dist_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(dim)))
            ws = torch.empty((max(nbytes, 16) + 3) // 4, dtype=torch.float32, device=dev)
            _C._check(lib.skgs_knn_dist_weights_backward(
                C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(dim), C.c_void_p(_C._ptr(pts)), C.c_void_p(_C._ptr(jts)),
                C.c_void_p(_C._ptr(rad)), C.c_void_p(_C._ptr(kw)), C.c_float(ctx.temperature), C.c_int32(0), C.c_int32(0),
                C.c_void_p(_C._ptr(w)),
                C.c_void_p(_C._ptr(idx)), C.c_void_p(_C._ptr(dist)), C.c_void_p(_C._ptr(g_w)), C.c_void_p(_C._ptr(g_p)),
                C.c_void_p(_C._ptr(g_j)), C.c_void_p(_C._ptr(g_r)), C.c_void_p(_C._ptr(g_k)), C.c_void_p(ws.data_ptr()),
                C.c_size_t(ws.numel() * 4), _C._stream()))
        return g_p, g_j, g_r, g_k, None, None


class _SpKnnWeights(torch.autograd.Function):
    """calc_LBS_weight over SUPERPOINT-sized tables (hundreds of bones; sk_gs.py:751-774 as stage ``sp`` calls it, :844) as the
    launches of ``FusedSuperpointStep``: ``skgs_sp_lbs_weights_forward`` (csrc/sp_knn.hip: 4 lanes per Gaussian, exact
    wave-wide prunes; 57 us at P = 100k, M = 512, 3 + 8 dimensions, where ``skgs_knn_dist_weights_forward`` -- built for the
    <= 60 bones of stage ``sk`` -- takes 705) with the weighting's parameters ACTIVATED as the reference passes them, and
    ``skgs_sp_lbs_weights_backward`` / ``skgs_lbs_weights_backward``.  The positions carry no gradient here (the caller
    detaches them, :753-755): gradients go to the hyper features, the radii / kernel weights, or the logit table."""

    @staticmethod
    def forward(ctx, xyz, feature, sp_xyz, sp_feature, kernel_radius, kernel_weight, sp_W, temperature: float, K: int):
        lib = _C.load_library()
        _C._require_gpu(xyz, 'points')
        dev = xyz.device
        f32 = lambda t: None if t is None else _C._f32c(t, dev)  # noqa: E731
        with _C._on_device(dev):
            pts, feat, sp, sfeat, rad, kw, spw = (f32(t) for t in (xyz, feature, sp_xyz, sp_feature, kernel_radius, kernel_weight, sp_W))
            P, M, F = pts.shape[0], sp.shape[0], 0 if feat is None else feat.shape[1]
            idx = torch.zeros((P, K), dtype=torch.int64, device=dev)
            w = torch.empty((P, K), dtype=torch.float32, device=dev)
            dist = torch.empty((P, K), dtype=torch.float32, device=dev)
            ptr = lambda t: C.c_void_p(_C._ptr(t))  # noqa: E731
            _C._check(lib.skgs_sp_lbs_weights_forward(
                C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), ptr(pts), ptr(feat), ptr(sp), ptr(sfeat), ptr(rad), ptr(kw),
                C.c_float(float(temperature)), ptr(spw), None, None, ptr(idx), ptr(w), ptr(dist), None, C.c_size_t(0), C.c_int32(1),
                C.c_int32(0), _C._stream()))
        ctx.save_for_backward(feat, sfeat, rad, kw, w, idx, dist)
        ctx.temperature, ctx.logits, ctx.M = float(temperature), spw is not None, M
        ctx.mark_non_differentiable(idx)
        return w, idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, _g_idx):
        feat, sfeat, rad, kw, w, idx, dist = ctx.saved_tensors
        lib = _C.load_library()
        dev = w.device
        (P, K), M = w.shape, ctx.M
        F = 0 if feat is None else feat.shape[1]
        need_f, need_sf, need_r, need_k, need_W = ctx.needs_input_grad[1], *ctx.needs_input_grad[3:7]
        ptr = lambda t: C.c_void_p(_C._ptr(t))  # noqa: E731
        g_f = g_sf = g_r = g_k = g_W = None
        with _C._on_device(dev):
            g_w = _C._f32c(g_w, dev)
            if ctx.logits:  # `W`: the weights do not depend on the distances
                if need_W:
                    g_W = torch.empty((P, M), dtype=torch.float32, device=dev)
                    _C._check(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(M), C.c_int32(K), ptr(w), ptr(idx), ptr(g_w), ptr(g_W),
                                                            _C._stream()))
            else:
                g_f = torch.empty((P, F), dtype=torch.float32, device=dev) if (need_f and F) else None
                g_sf = torch.empty((M, F), dtype=torch.float32, device=dev) if (need_sf and F) else None
                g_r = torch.empty((M,), dtype=torch.float32, device=dev) if (need_r and rad is not None) else None
                g_k = torch.empty((M,), dtype=torch.float32, device=dev) if (need_k and kw is not None) else None
                lib.skgs_sp_lbs_weights_workspace_bytes.restype = C.c_size_t
                ws = torch.empty((int(lib.skgs_sp_lbs_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(F))) + 16,),
                                 dtype=torch.uint8, device=dev)
                _C._check(lib.skgs_sp_lbs_weights_backward(
                    C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), ptr(feat), ptr(sfeat), ptr(rad), ptr(kw),
                    C.c_float(ctx.temperature), ptr(w), ptr(idx), ptr(dist), ptr(g_w), ptr(g_f), ptr(g_sf), ptr(g_r), ptr(g_k),
                    C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), C.c_int32(1), _C._stream()))
        return None, g_f, None, g_sf, g_r, g_k, g_W, None, None


def calc_lbs_weight(points: Tensor, joints: Tensor, K: int, sp_W: Optional[Tensor] = None,
                    kernel_radius: Optional[Tensor] = None, kernel_weight: Optional[Tensor] = None,
                    temperature: float = 1., feature: Optional[Tensor] = None, sp_feature: Optional[Tensor] = None
                    ) -> Tuple[Tensor, Tensor]:
    """``calc_LBS_weight`` of the reference (networks/sk_gs.py:751-774): K nearest bones + one of three weightings.
    Every branch is one HIP launch per direction (pytorch3d.knn_points semantics for the search): `W` on xyz through
    ``skgs_knn_lbs_weights``, the distance-based ones -- in 3 or 3 + 8 dimensions -- through ``skgs_knn_dist_weights_*``."""
    # superpoint-sized tables (stage sp): the search built for them -- when the positions carry no gradient (with hyper features
    # the reference detaches them, sk_gs.py:753-755), 0 or 8 hyper dimensions, and the table fits its LDS copy
    M = joints.shape[0]
    with_f = feature is not None and sp_feature is not None
    if (points.is_cuda and 60 < M <= 1024 and K <= 16 and K <= M and not (sp_W is not None and kernel_radius is not None)
            and (kernel_weight is None or kernel_radius is not None)
            and ((with_f and feature.shape[-1] == 8) or (not with_f and not points.requires_grad and not joints.requires_grad))):
        return _SpKnnWeights.apply(points.detach(), feature if with_f else None, joints.detach(), sp_feature if with_f else None,
                                   kernel_radius, kernel_weight, sp_W, float(temperature), K)
    if feature is not None and sp_feature is not None:
        points = torch.cat([points.detach(), feature], dim=-1)
        joints_q = torch.cat([joints.detach(), sp_feature], dim=-1)
    else:
        joints_q = joints
        # plain `W` method on xyz: search + gather + softmax in one launch (joint table in LDS: a few thousand bones at most)
        if sp_W is not None and kernel_radius is None and points.shape[-1] == 3 and K <= 16 and K <= sp_W.shape[1] <= 2048:
            return _KnnSoftmaxWeights.apply(points.detach(), joints.detach(), sp_W, K)
    if sp_W is None or kernel_radius is not None:
        # `weighted_kernel` / `kernel` / `dist` (sk_gs.py:759-766,770): search + weighting in one launch, and one launch
        # (+ a reduction of per-workgroup partials) for the gradients to the joints / features, radii and kernel weights
        if K <= 16 and points.shape[-1] <= 16:
            return _KnnDistWeights.apply(points, joints_q, kernel_radius, kernel_weight, float(temperature), K)
    with torch.no_grad():
        _, indices = _C.knn_bones(points.detach(), joints_q.detach(), K)
    if kernel_radius is not None or (sp_W is None):
        # (K > 16 or more than 16 dimensions: torch glue on [P,K]) distances must carry gradients to the joints
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
