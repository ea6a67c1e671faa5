"""Adaptive density control on top of ``FusedAdam`` -- the caller side of the hot path (scope row (f)-2 / (f)-4: the
optimizer-state surgery of ``change_optimizer`` and the clone / split / prune operations that use it).

Restated from ``GaussianSplatting`` (networks/gaussian_splatting.py): ``prune_points`` :565-576, ``densification_postfix``
:578-587, ``densify_and_split`` :589-620, ``densify_and_clone`` :622-634, ``densify`` :636-641, ``prune`` :643-650,
``reset_opacity`` :652-655.  The decisions (masks, samples) are a handful of torch ops; the surgery itself -- every
per-Gaussian parameter of the model (the reference's ``param_names_map`` plus the LBS logits ``sp_W``) pruned / extended
together with its two Adam moments -- is ONE row-gather launch per operation (``FusedAdam.gather_rows``,
csrc/densify.hip) instead of ~100 indexing / concatenation launches: clone + split + prune of 300k Gaussians take
well under a millisecond of device time.  After any of them the number of Gaussians has changed: rebuild ``FusedViewStep`` / flat gradient
buffers / captured graphs (their buffers are sized by P), as the reference re-creates its tensors.
"""
from typing import Dict, Optional

import torch
from torch import Tensor

import ctypes as C

from sk_gs_amd import _C
from sk_gs_amd.optim import FusedAdam

# module attribute -> optimizer group name (gaussian_splatting.py:89-96 + the per-Gaussian LBS logits)
PARAM_NAMES_MAP = {'_xyz': 'xyz', '_features_dc': 'f_dc', '_features_rest': 'f_rest', '_opacity': 'opacity',
                   '_scaling': 'scaling', '_rotation': 'rotation', 'sp_W': 'sp_W', 'hyper_feature': 'hyper'}  # (sk_gs.py:429)


class DensifyStats:
    """xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P] (gaussian_splatting.py:495-501); ``FusedViewStep`` has the
    same three attributes and can be passed wherever ``stats`` is expected."""

    def __init__(self, P: int, device):
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        self.max_radii2D = torch.zeros((P,), device=device)


def _names(model) -> Dict[str, str]:
    return {a: n for a, n in PARAM_NAMES_MAP.items() if getattr(model, a, None) is not None}


def quaternion_to_R(q: Tensor) -> Tensor:
    """xyzw quaternion (normalised first) -> rotation matrix, my_ext/ops_3d/quaternion.py:162-172"""
    x, y, z, w = torch.nn.functional.normalize(q, dim=-1).unbind(-1)
    return torch.stack([1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                        2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x,
                        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, 1 - 2 * x * x - 2 * y * y],
                       dim=-1).reshape(*x.shape, 3, 3)


def _rotate(R: Tensor, v: Tensor) -> Tensor:
    """``torch.bmm(R, v[..., None]).squeeze(-1)`` (gaussian_splatting.py:606) as three fused multiply-adds per row: the
    first bmm of a process initialises the BLAS library (~0.9 s on ROCm) for 3 x 3 products"""
    return (R * v[:, None, :]).sum(-1)


def _rebind(model, new: Dict[str, torch.nn.Parameter]):
    for attr, name in _names(model).items():
        if name in new and getattr(model, attr) is not new[name]:
            setattr(model, attr, new[name])
    model.P = model._xyz.shape[0]
    if getattr(model, 'capacity', None) is not None:  # in-place surgery: the kernels read the live count from this word
        model.capacity.set_live(model.P)


@torch.no_grad()
def prune_points(model, opt: FusedAdam, mask: Tensor, stats=None):
    """remove the Gaussians where ``mask`` is True (gaussian_splatting.py:565-576)"""
    keep = ~mask
    rows = torch.nonzero(keep).squeeze(1)
    _rebind(model, opt.gather_rows(list(_names(model).values()), rows, rows.numel()))
    if stats is not None:
        if hasattr(stats, 'gather_densify_stats'):
            stats.gather_densify_stats(rows)
        else:
            stats.xyz_gradient_accum = stats.xyz_gradient_accum[keep]
            stats.denom = stats.denom[keep]
            stats.max_radii2D = stats.max_radii2D[keep]


@torch.no_grad()
def densification_postfix(model, opt: FusedAdam, new_params: Dict[str, Tensor], stats=None):
    """append rows (by optimizer group name) and restart the statistics (gaussian_splatting.py:578-587)"""
    _rebind(model, opt.change_optimizer(new_params, op='concat'))
    _reset_stats(model, stats)


def _reset_stats(model, stats):
    if stats is not None and hasattr(stats, 'reset_densify_stats'):
        stats.reset_densify_stats()  # (FusedViewStep: in place -- a captured graph holds the addresses)
    elif stats is not None:
        P, dev = model._xyz.shape[0], model._xyz.device
        stats.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        stats.denom = torch.zeros((P, 1), device=dev)
        stats.max_radii2D = torch.zeros((P,), device=dev)


@torch.no_grad()
def densify_and_split(model, opt: FusedAdam, grads: Tensor, grad_threshold: float, scene_extent: float, N: int = 2,
                      stats=None, generator: Optional[torch.Generator] = None):
    """large Gaussians with a large screen-space gradient are replaced by N samples of themselves (:589-620)"""
    n_init = model._xyz.shape[0]
    padded = torch.zeros((n_init,), device=model._xyz.device)
    padded[:grads.shape[0]] = grads.squeeze()
    scaling = torch.exp(model._scaling)
    sel = (padded >= grad_threshold) & (scaling.amax(dim=1) > scene_extent)
    stds = scaling[sel].repeat(N, 1)
    samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
    rots = quaternion_to_R(model._rotation[sel]).repeat(N, 1, 1)
    new_xyz = _rotate(rots, samples) + model._xyz[sel].repeat(N, 1)
    new_scaling = torch.log(scaling[sel].repeat(N, 1) / (0.8 * N))
    # densification_postfix (append N copies of the selected rows) followed by prune_points (drop the selected originals,
    # :619-620) as ONE gather: the unselected rows keep parameters and moments, the N x S new rows follow with zero moments
    keep_rows, sel_rows = torch.nonzero(~sel).squeeze(1), torch.nonzero(sel).squeeze(1)
    n_keep = keep_rows.numel()
    _rebind(model, opt.gather_rows(list(_names(model).values()), torch.cat([keep_rows, sel_rows.repeat(N)]), n_keep))
    model._xyz.data[n_keep:] = new_xyz
    model._scaling.data[n_keep:] = new_scaling
    _reset_stats(model, stats)


@torch.no_grad()
def densify_and_clone(model, opt: FusedAdam, grads: Tensor, grad_threshold: float, scene_extent: float, stats=None):
    """small Gaussians with a large screen-space gradient are duplicated (:622-634)"""
    sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (torch.exp(model._scaling).amax(dim=1) <= scene_extent)
    P = model._xyz.shape[0]
    rows = torch.cat([torch.arange(P, device=sel.device), torch.nonzero(sel).squeeze(1)])
    _rebind(model, opt.gather_rows(list(_names(model).values()), rows, P))
    _reset_stats(model, stats)


@torch.no_grad()
def densify(model, opt: FusedAdam, stats, max_grad: float, extent: float, densify_percent_dense: float = 0.01,
            generator: Optional[torch.Generator] = None, N: int = 2):
    """clone, then split, by the mean screen-space gradient accumulated in ``stats`` (:636-641) -- as ONE row gather.
    The two selections are disjoint (clone: max scale <= percent_dense * extent, split: > it) and a fresh clone carries a
    zero gradient (the padded ``grads`` of :591-593), so ``densify_and_clone`` followed by ``densify_and_split`` leaves
        [originals not selected for splitting] + [clones] + [N samples of every split Gaussian]
    with the moments of the first group kept and the others zero: exactly one gather (same rows, same order, same random
    samples as the two calls)."""
    if not model._xyz.is_cuda:
        raise _C.SkgsError('densify needs the model on a HIP device')
    # decisions and row list on the device (csrc/densify.hip): one flag launch, one compaction launch, ONE read-back of the
    # three group sizes (the ~25 torch kernels and three boolean-index synchronisations of the torch formulation were
    # most of a densification: 1.6 ms at 300k Gaussians, 0.4 now)
    lib = _C.load_library()
    P, dev = model._xyz.shape[0], model._xyz.device
    rows = torch.empty(((2 + N) * P,), dtype=torch.int64, device=dev)
    counts = torch.empty((3,), dtype=torch.int32, device=dev)
    lib.skgs_select_workspace_bytes.restype = C.c_size_t
    flags = torch.empty((int(lib.skgs_select_workspace_bytes(C.c_int32(P))),), dtype=torch.uint8, device=dev)
    acc, den = stats.xyz_gradient_accum, stats.denom
    assert acc.is_contiguous() and den.is_contiguous() and acc.numel() == P and model._scaling.is_contiguous()
    _C._check(lib.skgs_densify_select(
        C.c_int32(P), C.c_void_p(acc.data_ptr()), C.c_void_p(den.data_ptr()), C.c_void_p(model._scaling.data_ptr()),
        C.c_float(max_grad), C.c_float(densify_percent_dense * extent), C.c_int32(N), C.c_void_p(rows.data_ptr()),
        C.c_void_p(counts.data_ptr()), C.c_void_p(flags.data_ptr()), _C._stream()))
    n_keep, n_clone, n_split = counts.tolist()  # (synchronises)
    n_new = N * n_split
    _rebind(model, opt.gather_rows(list(_names(model).values()), rows[:n_keep + n_clone + n_new], n_keep))
    if n_new:  # the children are copies of their parents: sample position and shrink scale in place
        gdev = generator.device if generator is not None else dev  # (a CPU generator draws on the host)
        normals = torch.randn((n_new, 3), device=gdev, generator=generator).to(dev)
        _C._check(lib.skgs_split_children(
            C.c_int32(n_new), C.c_int32(N), C.c_void_p(normals.data_ptr()), C.c_void_p(model._xyz.data[-n_new:].data_ptr()),
            C.c_void_p(model._scaling.data[-n_new:].data_ptr()), C.c_void_p(model._rotation.data[-n_new:].data_ptr()),
            _C._stream()))
    _reset_stats(model, stats)


@torch.no_grad()
def prune(model, opt: FusedAdam, stats, min_opacity: float, extent: float, max_screen_size: float,
          prune_percent_dense: float = 0.1):
    """drop transparent, screen-filling and world-size outliers (:643-650)"""
    lib = _C.load_library()
    P, dev = model._xyz.shape[0], model._xyz.device
    rows = torch.empty((P,), dtype=torch.int64, device=dev)
    counts = torch.empty((1,), dtype=torch.int32, device=dev)
    lib.skgs_select_workspace_bytes.restype = C.c_size_t
    flags = torch.empty((int(lib.skgs_select_workspace_bytes(C.c_int32(P))),), dtype=torch.uint8, device=dev)
    radii = stats.max_radii2D if max_screen_size else None
    assert model._opacity.is_contiguous() and model._scaling.is_contiguous() and (radii is None or radii.is_contiguous())
    _C._check(lib.skgs_prune_select(
        C.c_int32(P), C.c_void_p(model._opacity.data_ptr()), C.c_void_p(None if radii is None else radii.data_ptr()),
        C.c_void_p(model._scaling.data_ptr()), C.c_float(min_opacity), C.c_float(max_screen_size or 0.0),
        C.c_float(prune_percent_dense * extent), C.c_void_p(rows.data_ptr()), C.c_void_p(counts.data_ptr()),
        C.c_void_p(flags.data_ptr()), _C._stream()))
    n_keep = int(counts.item())  # (synchronises)
    rows = rows[:n_keep]
    _rebind(model, opt.gather_rows(list(_names(model).values()), rows, n_keep))
    if stats is not None and hasattr(stats, 'gather_densify_stats'):
        stats.gather_densify_stats(rows)
    elif stats is not None:
        stats.xyz_gradient_accum = stats.xyz_gradient_accum.index_select(0, rows)
        stats.denom = stats.denom.index_select(0, rows)
        stats.max_radii2D = stats.max_radii2D.index_select(0, rows)


@torch.no_grad()
def reset_opacity(model, opt: FusedAdam):
    """opacity <- min(opacity, 0.01) in logit space, with fresh Adam moments (:652-655)"""
    o = torch.minimum(torch.sigmoid(model._opacity), torch.full_like(model._opacity, 0.01))
    _rebind(model, opt.change_optimizer(torch.log(o / (1 - o)), name='opacity', op='replace'))


# ------------------------------------------------------------------------------------------------ spatial order
def morton_order(xyz: Tensor, bits: int = 10) -> Tensor:
    """row permutation that sorts the positions along a Z-order curve (``bits`` per axis over their bounding box): Gaussians
    that are neighbours in space become neighbours in memory -- and in a wavefront"""
    lo, hi = xyz.min(0).values, xyz.max(0).values
    q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


@torch.no_grad()
def sort_spatially(model, opt: Optional[FusedAdam] = None, stats=None) -> Tensor:
    """Re-order the Gaussians along a Z-order curve: every per-Gaussian parameter, its Adam moments (ONE row-gather launch,
    ``FusedAdam.gather_rows``) and the densification statistics move together, so training is unaffected -- the order of the
    Gaussians carries no meaning (the reference appends clones and split children at the end, gaussian_splatting.py:577-587).
    What it buys: the lanes of a wavefront then hold spatial neighbours, which share their nearest bones / superpoints and
    their screen tiles -- the wave-wide early-outs of the superpoint search (csrc/sp_knn.hip) skip most candidates, and the
    bone-gradient accumulation of the skinning backward merges a wave's contributions before it touches LDS.  Call it after
    a densification event (the same cadence as the reference's surgery); returns the permutation applied."""
    rows = morton_order(model._xyz.detach())
    if opt is None:
        for attr in _names(model):
            p = getattr(model, attr)
            p.data = p.data.index_select(0, rows).contiguous()
    else:
        _rebind(model, opt.gather_rows(list(_names(model).values()), rows, rows.numel()))
    if stats is not None:
        if hasattr(stats, 'gather_densify_stats'):
            stats.gather_densify_stats(rows)
        else:
            stats.xyz_gradient_accum = stats.xyz_gradient_accum[rows]
            stats.denom, stats.max_radii2D = stats.denom[rows], stats.max_radii2D[rows]
        if getattr(stats, 'sparse_logits', False) and opt is not None:  # stage sp, `W`: the logit table's rows moved with the rest
            stats.refresh_logit_mask(opt)
    return rows
