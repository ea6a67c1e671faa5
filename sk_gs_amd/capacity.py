"""Room to grow: per-Gaussian storage with a capacity, so that densification does not invalidate a captured step.

The reference re-creates every per-Gaussian tensor at each clone / split / prune (``change_optimizer`` /
``densification_postfix`` / ``prune_points``, networks/gaussian_splatting.py:515-587): new storage, new optimizer state,
new shapes.  A hipGraph of the training step bakes pointers, grids and the Gaussian count into its kernel nodes, so
rounds 1-2 re-built the step's buffers and re-captured the graph after every densification: ~1 ms of surgery + 0.5 ms of
rebuild + 2 ms of capture every 100 steps, 9 % of the training time at a 0.39 ms step.

With a ``RowCapacity`` every per-Gaussian parameter (and, through it, its gradient slot in the flat buffer, its two Adam
moments and the step's per-Gaussian workspaces) lives in storage of ``P_cap`` rows of which the first ``P`` are live.
The kernels are launched for ``P_cap`` rows and read the live count from ONE device word
(``skgs_raster_inputs.live_count`` / ``skgs_deform_inputs.live_count``, include/skgs.h); the optimizer's descriptor table
-- device memory -- carries the live element counts.  Clone / split / prune then gather the rows through a scratch copy
back INTO the same storage, refresh the Python-side views and the two device-side facts (live count, table): nothing the
captured graph baked in has moved, and it keeps replaying.  Growing beyond the capacity falls back to a rebuild.
"""
from typing import Optional

import torch
from torch import Tensor

# module attributes of SkinnedGaussians with one row per Gaussian (sk_gs_amd.densify.PARAM_NAMES_MAP)
PER_GAUSSIAN = ('_xyz', '_features_dc', '_features_rest', '_opacity', '_scaling', '_rotation', 'sp_W', 'hyper_feature')


def cap_store(p: Tensor) -> Optional[Tensor]:
    """the capacity storage ``[P_cap, ...]`` behind a per-Gaussian parameter, or None"""
    return getattr(p, '_cap_store', None)


def slot_numel(p: Tensor) -> int:
    """elements to reserve for ``p`` wherever something mirrors it (gradient slot, Adam moments, chunk table)"""
    s = cap_store(p)
    return p.numel() if s is None else s.numel()


def regrad(p: Tensor) -> None:
    """re-shape ``p.grad`` to ``p``'s (new) row count over the SAME storage: the slot was reserved with ``slot_numel``"""
    g = p.grad
    if g is None or tuple(g.shape) == tuple(p.shape):
        return
    room = getattr(p, '_grad_slot', 0)
    assert p.numel() <= room, f'gradient slot of {room} elements cannot hold {p.numel()}: allocate it with slot_numel(p)'
    strides = []
    acc = 1
    for d in reversed(p.shape):
        strides.insert(0, acc)
        acc *= d
    p.grad = torch.as_strided(g, tuple(p.shape), tuple(strides), g.storage_offset())


class RowCapacity:
    def __init__(self, model, P_cap: int):
        P = int(model.P)
        assert P_cap >= P
        self.P_cap = int(P_cap)
        dev = model._xyz.device
        assert model._xyz.is_cuda, 'a row capacity is a device-side contract (the kernels read the live count)'
        self.live = torch.tensor([P], dtype=torch.int32, device=dev)  # THE device word the kernels read
        with torch.no_grad():
            for attr in PER_GAUSSIAN:
                p = getattr(model, attr, None)
                if p is None:
                    continue
                store = torch.zeros((self.P_cap,) + tuple(p.shape[1:]), dtype=p.dtype, device=dev)
                store[:P].copy_(p.data)
                p.data = store[:P]
                p._cap_store = store
                if p.grad is not None:  # a gradient allocated for P rows has no room: start over
                    p.grad = None

    @torch.no_grad()
    def grow(self, model, P_cap: int, optimizer=None):
        """Re-home the per-Gaussian parameters (and, with ``optimizer``, their Adam moments) into storage of ``P_cap`` rows --
        what to do on ``CapacityExceeded``.  Every address changes: afterwards rebuild the gradient buffer (``FlatGradBuffer``)
        and the step (``FusedViewStep``), ``optimizer.rebind()``, and let the graphs be captured again -- the cost of ONE
        round-2 densification, paid once per capacity doubling instead of at every event."""
        assert P_cap >= self.P_cap
        P = int(model.P)
        for attr in PER_GAUSSIAN:
            p = getattr(model, attr, None)
            if p is None:
                continue
            store = torch.zeros((int(P_cap),) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
            store[:P].copy_(p.data)
            p.data = store[:P]
            p._cap_store = store
            p.grad = None  # (its slot was sized for the old capacity)
            if optimizer is not None and p in optimizer.state:
                st = optimizer.state[p]
                m, v = torch.zeros_like(store), torch.zeros_like(store)
                m[:P].copy_(st['exp_avg']), v[:P].copy_(st['exp_avg_sq'])
                optimizer._cap_state[p] = (m, v)
                optimizer.state[p] = dict(exp_avg=m[:P], exp_avg_sq=v[:P])
        self.P_cap = int(P_cap)
        self.live.fill_(P)  # (the same device word: a NEW step picks it up again)

    def set_live(self, n: int):
        assert 0 <= n <= self.P_cap
        self.live.fill_(int(n))

    def fits(self, n: int) -> bool:
        return n <= self.P_cap
