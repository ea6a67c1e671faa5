"""The two regularisers of stage ``sp`` on the LBS weights as one launch each (value AND gradient): ``skgs_weight_sparsity`` /
``skgs_weight_smooth`` (csrc/weight_reg.hip) behind torch's autograd API.

The reference (networks/sk_gs.py:1339-1340, 1357-1359; weights 0.1 each in exps/default.yaml:85-86, applied to ``outputs['_knn_w']`` in every
``sp`` iteration, :1572-1574)::

    loss_weight_sparsity(w, eps=1e-7) = -(w * log(w + eps) + (1 - w) * log(1 - w + eps)).mean()
    loss_weight_smooth(w)             = (w[:, None] - w[gs_knn_index]).abs().mean()          # gs_knn_index [P, 21]: a [P, 21, K] gather

``accelerate_reference()`` patches the two methods with these (same arguments, same value, same gradient; any call outside the
conditions -- a CPU tensor, more than 16 neighbours per row, an index table of another size -- reaches the reference's own lines)."""
import ctypes as C

import torch

from sk_gs_amd import _C


def _p(t):
    return C.c_void_p(None if t is None else t.data_ptr())


def _partials(dev):
    lib = _C.load_library()
    lib.skgs_weight_reg_partials.restype = C.c_int32
    return torch.empty(int(lib.skgs_weight_reg_partials()), dtype=torch.float32, device=dev)


class _Sparsity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, eps):
        wc = w.detach().contiguous()
        grad, part = torch.empty_like(wc), _partials(wc.device)
        _C._check(_C.load_library().skgs_weight_sparsity(C.c_int64(wc.numel()), _p(wc), C.c_float(eps), _p(grad), _p(part), _C._stream()))
        ctx.save_for_backward(grad)
        ctx.shape = w.shape
        return part.sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None


_inverse = {}    # id(neighbour table) -> (weak reference to it, its version, offsets [P + 1] int32, sources int32)


def inverse_lists(nbr):
    """for every Gaussian j the Gaussians that list it in ``nbr`` [P, G] (CSR, int32), cached per table object and version: the reference
    rebuilds ``gs_knn_index`` every 1000-3000 iterations (sk_gs.py:1342-1355), the lists follow"""
    import weakref
    hit = _inverse.get(id(nbr))
    if hit is None or hit[0]() is not nbr or hit[1] != nbr._version:
        for k in [k for k, v in _inverse.items() if v[0]() is None]:
            del _inverse[k]
        P, G = nbr.shape
        with torch.no_grad():
            flat = nbr.reshape(-1)
            flat = torch.where(flat < 0, flat + P, flat)
            ok = (flat >= 0) & (flat < P)
            src = torch.arange(P, device=nbr.device).repeat_interleave(G)[ok]
            dst = flat[ok]
            order = torch.argsort(dst, stable=True)
            offsets = torch.zeros(P + 1, dtype=torch.int64, device=nbr.device)
            offsets[1:] = torch.cumsum(torch.bincount(dst, minlength=P), 0)
            hit = _inverse[id(nbr)] = (weakref.ref(nbr), nbr._version, offsets.to(torch.int32).contiguous(), src[order].to(torch.int32).contiguous())
    return hit[2], hit[3]


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, nbr):
        wc = w.detach().contiguous()
        P, K = wc.shape
        grad, part = torch.empty_like(wc), _partials(wc.device)
        off, src = inverse_lists(nbr)
        _C._check(_C.load_library().skgs_weight_smooth(C.c_int32(P), C.c_int32(K), C.c_int32(nbr.shape[1]), _p(wc), _p(nbr), _p(off), _p(src),
                                                      _p(grad), _p(part), _C._stream()))
        ctx.save_for_backward(grad)
        return part.sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def sparsity_supported(w) -> bool:
    return torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.numel() > 0


def smooth_supported(w, nbr) -> bool:
    return (torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and 1 <= w.shape[1] <= 16 and torch.is_tensor(nbr)
            and nbr.is_cuda and nbr.dtype == torch.int64 and nbr.dim() == 2 and nbr.shape[0] == w.shape[0] and nbr.is_contiguous() and w.shape[0] > 0)


def weight_sparsity(w, eps: float = 1e-7):
    """``-(w log(w + eps) + (1 - w) log(1 - w + eps)).mean()`` of a HIP fp32 tensor: one launch forward, one multiply backward"""
    return _Sparsity.apply(w, float(eps))


def weight_smooth(w, neighbours):
    """``(w[:, None] - w[neighbours]).abs().mean()`` for w [P, K], neighbours [P, G] int64 on a HIP device"""
    return _Smooth.apply(w, neighbours)
