"""Stand-in for ``pytorch3d.ops`` (facebookresearch/pytorch3d, un-pinned in the reference's requirements_git.txt:3) -- what the
reference imports from it, so that the UNMODIFIED ``networks/sk_gs.py`` runs on a machine without that CUDA-only package::

    import sk_gs_amd
    sk_gs_amd.install_as_pytorch3d()          # sys.modules['pytorch3d'], ['pytorch3d.ops'];  `from pytorch3d.ops import knn_points`

Call sites: ``knn_points(points[None], sp_points[None], None, None, K=K)`` in calc_LBS_weight (networks/sk_gs.py:757, the per-frame
search of the deform: 100k Gaussians against 20 joints in 3 dimensions, or against 512 superpoints in 3 + 8), :1365 (Gaussian-to-
Gaussian neighbours for the regularisers), networks/losses/SC_GS_arap_loss.py:25,61,74 and ``ball_query`` there (:210).

Semantics restated from pytorch3d's published interface (its source is not in the reference tree: parity unpinned):
``knn_points(p1 [N,P1,D], p2 [N,P2,D], lengths1, lengths2, norm, K, version, return_nn, return_sorted)`` -> namedtuple
``(dists [N,P1,K], idx [N,P1,K] int64, knn [N,P1,K,D] | None)``; dists are SQUARED L2 (norm = 2; L1 for norm = 1), ascending;
rows with fewer than K valid neighbours are padded with 0; equal distances keep the lower index first (this package's rule, the
same as csrc/deform.hip and the oracle); ``dists`` is differentiable w.r.t. p1 and p2 (d = sum (p1 - p2[idx])^norm).

On a HIP device the per-frame call shape (one cloud, no lengths, squared L2, K <= 16, D <= 16, fp32) is ONE launch of
libskgs_hip.so: ``skgs_sp_lbs_weights_forward`` for superpoint-sized tables (60 < M <= 1024 in 3 or 3 + 8 dimensions: the
wave-cooperative pruned scan of csrc/sp_knn.hip) and ``skgs_knn_bones`` otherwise -- no fallback when the library is missing.
Everything else (CPU tensors, batches, lengths, L1, large K) runs as chunked pure torch.  The indices that call returns are typed
(``NeighbourIndex``, a ``torch.Tensor`` subclass over the same int64 storage) so that the reference's own gathers of per-bone rows by
them -- ``sk_d_rot[indices]``, ``kernel_radius[indices]`` ... -- get a backward (``skgs_index_add_rows``) that does not walk
duplicates serially: torch's index backward is 7.5 ms per gather at 100k x 5 indices into 20 rows (``SKGS_KNN_TYPED_INDEX=0``: plain).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import namedtuple
from typing import Optional, Union

import torch
from torch import Tensor

__all__ = ['knn_points', 'knn_gather', 'ball_query']

_KNN = namedtuple('KNN', 'dists idx knn')
_TYPED_INDEX = os.environ.get('SKGS_KNN_TYPED_INDEX', '1') != '0'
hip_calls = {'knn_bones': 0, 'sp_search': 0, 'gather_backward': 0}  # counters (tests)


def _index_add_rows(index: Tensor, rows: Tensor, shape) -> Tensor:
    """``zeros(shape).index_add_(0, index.flatten(), rows.reshape(-1, *shape[1:]))`` on a HIP device: skgs_index_add_rows"""
    from sk_gs_amd import _C
    M, Cn = shape[0], 1
    for d in shape[1:]:
        Cn *= d
    lib = _C.load_library()
    dev = rows.device
    with _C._on_device(dev):
        g = _C._f32c(rows, dev)
        idx = index.contiguous()
        R = idx.numel()
        out = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
        lib.skgs_index_add_rows_workspace_bytes.restype = C.c_size_t
        nbytes = int(lib.skgs_index_add_rows_workspace_bytes(C.c_int64(R), C.c_int32(Cn), C.c_int32(M)))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        _C._check(lib.skgs_index_add_rows(C.c_int64(R), C.c_int32(Cn), C.c_int32(M), C.c_void_p(_C._ptr(idx)), C.c_void_p(_C._ptr(g)),
                                          C.c_void_p(_C._ptr(out)), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel() * 4), _C._stream()))
    hip_calls['gather_backward'] += 1
    return out


class _GatherRows(torch.autograd.Function):
    """``table[index]`` for a per-bone table [M, ...] and neighbour indices [P, K]: torch's gather forward; backward = the rows summed
    per bone by ``skgs_index_add_rows`` (csrc/lie_blend.hip) instead of torch's sort-and-walk index backward, whose cost grows with the
    number of DUPLICATES per row -- 100k x 5 indices into 20 bones: 7.5 ms per gather on an MI355X, four of them per step of the
    reference's stage `sk` (sk_gs.py:1148-1149 `sk_d_rot[indices]`, `sk_d_scale[indices]`; :760-763 `kernel_radius[indices]`)."""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.table_shape = table.shape
        return table[index]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        (index,) = ctx.saved_tensors
        return _index_add_rows(index, grad, ctx.table_shape), None


class NeighbourIndex(torch.Tensor):
    """What ``knn_points`` returns as ``idx`` on a HIP device: the int64 indices (a real tensor, same storage) typed so that the
    reference's ``table[indices]`` gathers of per-bone rows (see ``_GatherRows``) get a backward that does not degrade with duplicates.
    Only that one expression is changed; slicing / reshaping the indices keeps the type, every other use -- ``torch.gather(sp_W, 1,
    indices)``, ``SE3[indices]``, ``indices.detach()`` (what the reference stores in its buffers, sk_gs.py:772-773), comparisons,
    arithmetic -- sees and returns plain tensors."""
    _KEEP = None

    @classmethod
    def wrap(cls, idx: Tensor) -> Tensor:
        return torch.Tensor._make_subclass(cls, idx)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if cls._KEEP is None:
            T = torch.Tensor
            cls._KEEP = {T.__getitem__, T.select, T.view, T.reshape, T.contiguous, T.squeeze, T.unsqueeze, T.expand, T.flatten, T.clone}
        with torch._C.DisableTorchFunctionSubclass():
            if (func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], cls) and not isinstance(args[0], cls)
                    and isinstance(args[0], Tensor) and args[0].dtype == torch.float32 and args[0].is_cuda and args[0].requires_grad
                    and torch.is_grad_enabled() and args[0].dim() >= 1 and args[0].shape[0] <= (1 << 20)):
                return _GatherRows.apply(args[0], args[1].as_subclass(Tensor))
            out = func(*args, **kwargs)
            if func in cls._KEEP and len(args) and isinstance(args[0], cls) and isinstance(out, Tensor) and out.dtype == torch.int64:
                return out.as_subclass(cls)
            return out


def _pairwise(p1: Tensor, p2: Tensor, norm: int) -> Tensor:
    """[P1,P2] distances accumulated over the coordinates in index order (what a per-thread loop over D does)"""
    d = None
    for c in range(p1.shape[-1]):
        diff = p1[:, None, c] - p2[None, :, c]
        term = diff * diff if norm == 2 else diff.abs()
        d = term if d is None else d + term
    return d


def _search_torch(p1: Tensor, p2: Tensor, K: int, norm: int):
    """indices [P1,K] of the K nearest rows of p2 (ascending distance, ties to the lower index), chunked over P1"""
    P1, P2 = p1.shape[0], p2.shape[0]
    k = min(K, P2)
    out = torch.zeros((P1, K), dtype=torch.int64, device=p1.device)
    if P1 == 0 or k == 0:
        return out
    chunk = max(1, (1 << 24) // max(P2, 1))
    for a in range(0, P1, chunk):
        d = _pairwise(p1[a:a + chunk], p2, norm)
        order = torch.sort(d, dim=1, stable=True).indices[:, :k]
        out[a:a + chunk, :k] = order
    return out


class _KnnHip(torch.autograd.Function):
    """search on the HIP device; distances differentiable w.r.t. both clouds (pytorch3d's _knn_points.backward:
    grad_p1 = 2 g (p1 - p2[idx]) summed over k, grad_p2 the scatter of its negative)"""

    @staticmethod
    def forward(ctx, p1, p2, K: int):
        from sk_gs_amd import _C
        lib = _C.load_library()
        dev = p1.device
        with _C._on_device(dev):
            a, b = _C._f32c(p1, dev), _C._f32c(p2, dev)
            (P, D), M = a.shape, b.shape[0]
            dist = torch.empty((P, K), dtype=torch.float32, device=dev)
            idx = torch.empty((P, K), dtype=torch.int64, device=dev)
            ptr = lambda t: C.c_void_p(_C._ptr(t))  # noqa: E731
            if 60 < M <= 1024 and D in (3, 11):
                F = D - 3
                xyz, feat = (a, None) if F == 0 else (a[:, :3].contiguous(), a[:, 3:].contiguous())
                sxyz, sfeat = (b, None) if F == 0 else (b[:, :3].contiguous(), b[:, 3:].contiguous())
                w = torch.empty((P, K), dtype=torch.float32, device=dev)  # the launch also weights the neighbours (`dist` rule): unused
                _C._check(lib.skgs_sp_lbs_weights_forward(
                    C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(F), ptr(xyz), ptr(feat), ptr(sxyz), ptr(sfeat), None, None,
                    C.c_float(1.0), None, None, None, ptr(idx), ptr(w), ptr(dist), None, C.c_size_t(0), C.c_int32(1), C.c_int32(0),
                    _C._stream()))
                hip_calls['sp_search'] += 1
            else:
                _C._check(lib.skgs_knn_bones(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(D), ptr(a), ptr(b), ptr(dist), ptr(idx),
                                             _C._stream()))
                hip_calls['knn_bones'] += 1
        ctx.save_for_backward(a, b, idx)
        ctx.mark_non_differentiable(idx)
        return dist, idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_dist, _g_idx):
        a, b, idx = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[:2]
        g_a = g_b = None
        if need_a or need_b:
            t = (2.0 * g_dist).unsqueeze(-1) * (a[:, None, :] - b[idx])  # [P,K,D]
            if need_a:
                g_a = t.sum(dim=1)
            if need_b:
                g_b = _index_add_rows(idx, -t, b.shape)
        return g_a, g_b, None


def knn_points(p1: Tensor, p2: Tensor, lengths1: Optional[Tensor] = None, lengths2: Optional[Tensor] = None, norm: int = 2,
               K: int = 1, version: int = -1, return_nn: bool = False, return_sorted: bool = True) -> _KNN:
    """K nearest neighbours in ``p2`` of every point of ``p1`` (see the module docstring)."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError('pts1 and pts2 must have the same batch dimension.')
    if p1.shape[2] != p2.shape[2]:
        raise ValueError('pts1 and pts2 must have the same point dimension.')
    if norm not in (1, 2):
        raise ValueError('Support for 1 or 2 norm.')
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    if (p1.is_cuda and N == 1 and lengths1 is None and lengths2 is None and norm == 2 and 1 <= K <= min(16, P2) and D <= 16
            and p1.dtype == torch.float32 and p2.dtype == torch.float32 and P1 > 0):
        dists, idx = _KnnHip.apply(p1[0], p2[0], K)
        dists, idx = dists[None], (NeighbourIndex.wrap(idx[None]) if _TYPED_INDEX else idx[None])
    else:
        p1c, p2c = p1.contiguous(), p2.contiguous()
        idx_rows, dist_rows = [], []
        for n in range(N):
            n1 = P1 if lengths1 is None else int(lengths1[n])
            n2 = P2 if lengths2 is None else int(lengths2[n])
            with torch.no_grad():
                idx_n = torch.zeros((P1, K), dtype=torch.int64, device=p1.device)
                idx_n[:n1] = _search_torch(p1c[n, :n1], p2c[n, :n2], K, norm)
            diff = p1c[n, :, None, :] - p2c[n][idx_n]
            d_n = (diff * diff if norm == 2 else diff.abs()).sum(dim=-1)
            valid = torch.zeros((P1, K), dtype=torch.bool, device=p1.device)
            valid[:n1, :min(K, n2)] = True
            dist_rows.append(torch.where(valid, d_n, torch.zeros_like(d_n)))
            idx_rows.append(torch.where(valid, idx_n, torch.zeros_like(idx_n)))
        dists, idx = torch.stack(dist_rows), torch.stack(idx_rows)
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def knn_gather(x: Tensor, idx: Tensor, lengths: Union[Tensor, None] = None) -> Tensor:
    """``x [N,M,U]``, ``idx [N,L,K]`` -> ``[N,L,K,U]`` with ``out[n,l,k] = x[n, idx[n,l,k]]`` (rows beyond ``lengths`` are 0)"""
    N, M, U = x.shape
    _N, L, K = idx.shape
    if N != _N:
        raise ValueError('x and idx must have same batch dimension.')
    if lengths is None:
        lengths = torch.full((N,), M, dtype=torch.int64, device=x.device)
    idx_e = idx[:, :, :, None].expand(-1, -1, -1, U)
    out = x[:, :, None].expand(-1, -1, K, -1).gather(1, idx_e)
    needs_mask = lengths.min() < K
    if needs_mask:
        mask = lengths[:, None] <= torch.arange(K, device=x.device)[None]
        mask = mask[:, None].expand(-1, L, -1)[:, :, :, None].expand(-1, -1, -1, U)
        out = out.masked_fill(mask, 0.0)
    return out


def ball_query(p1: Tensor, p2: Tensor, lengths1: Optional[Tensor] = None, lengths2: Optional[Tensor] = None, K: int = 500,
               radius: float = 0.2, return_nn: bool = True) -> _KNN:
    """the first (in index order) up to K points of ``p2`` within ``radius`` of each point of ``p1``: dists squared, idx padded
    with -1 (networks/losses/SC_GS_arap_loss.py:210).  Pure torch (not on the per-frame path)."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError('pts1 and pts2 must have the same batch dimension.')
    if p1.shape[2] != p2.shape[2]:
        raise ValueError('pts1 and pts2 must have the same point dimension.')
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    idx = torch.full((N, P1, K), -1, dtype=torch.int64, device=p1.device)
    r2 = radius * radius
    with torch.no_grad():
        for n in range(N):
            n1 = P1 if lengths1 is None else int(lengths1[n])
            n2 = P2 if lengths2 is None else int(lengths2[n])
            chunk = max(1, (1 << 24) // max(n2, 1))
            for a in range(0, n1, chunk):
                inside = _pairwise(p1[n, a:a + chunk], p2[n, :n2], 2) < r2
                rank = inside.cumsum(dim=1) - 1                         # position of each hit among its row's hits
                take = inside & (rank < K)
                rows, cols = take.nonzero(as_tuple=True)
                idx[n, a + rows, rank[rows, cols]] = cols
    safe = idx.clamp(min=0)
    gathered = p2[torch.arange(N, device=p1.device)[:, None, None], safe]
    diff = p1[:, :, None, :] - gathered
    dists = torch.where(idx >= 0, (diff * diff).sum(dim=-1), torch.zeros((), dtype=p1.dtype, device=p1.device))
    nn = torch.where((idx >= 0)[..., None], gathered, torch.zeros((), dtype=p1.dtype, device=p1.device)) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)
