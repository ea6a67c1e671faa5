"""View-parallel training: one process per GPU, every rank holds a full replica of the Gaussians, renders a different
training view, and the parameter gradients are summed with ONE all-reduce over RCCL/xGMI.

The reference has no working multi-GPU path for this model (its DDP wrapper breaks on the first densification,
my_ext/framework.py:339-357; scripts/run_all_dnerf.sh runs one scene per GPU), so this is new work, not a port:
  * all parameter gradients live in ONE flat fp32 buffer (``FlatGradBuffer``): ``p.grad`` are views into it, so a
    single collective moves everything -- on the 8-GPU fully connected xGMI box RCCL splits a large buffer over all
    7 links, which per-tensor all-reduces of the small tensors (opacity, rotation, ...) would not;
  * densification statistics (sum of screen-space gradient norms, visit counts: SUM; max radii: MAX,
    gaussian_splatting.py:503-513) are reduced the same way so prune/split decisions stay identical on every rank.
Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` on CPU (tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from sk_gs_amd.capacity import slot_numel


def init_distributed(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, world, local_rank) from the torchrun environment; initialises the default group when world > 1
    (or, with ``force``, even for a single rank: lets one GPU exercise the collective code path)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test knobs: SKGS_SHARE_GPU=1 lets several ranks use one device (RCCL refuses that, so together with
    # SKGS_DIST_BACKEND=gloo): a 1-GPU box can then run the real multi-rank schedule (tests/test_gpu_bench_contract.py)
    if os.environ.get('SKGS_SHARE_GPU') and torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()
    backend = os.environ.get('SKGS_DIST_BACKEND') or backend
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))  # barrier() knows its device
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class FlatGradBuffer:
    """All gradients of ``params`` as views of one contiguous buffer.  Every view starts on a 16-byte boundary (an 11-float
    bias in the middle would otherwise leave every later tensor off the float4 grid: the Adam kernel then takes its scalar
    path for them, 13 us instead of 4 for the deform network's weights); the padding words stay zero."""

    ALIGN = 4  # floats

    @classmethod
    def _offsets(cls, params):
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (slot_numel(p) + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN  # (a row capacity reserves the whole capacity)
        return offs, off

    def __init__(self, params: Iterable[Tensor]):
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        self._offs, total = self._offsets(self.params)
        dev = self.params[0].device if self.params else 'cpu'
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self._offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            p._grad_slot = slot_numel(p)

    def zero_(self):
        self.flat.zero_()

    def rebind(self):
        """re-attach the views (needed if something replaced p.grad, e.g. zero_grad(set_to_none=True))"""
        for p, off in zip(self.params, self._offs):
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + n].data_ptr() or p.grad.shape != p.shape:
                p.grad = self.flat[off:off + n].view_as(p)

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


class ViewParallel:
    """Gradient (and densification-statistic) reduction over the ranks of the default process group."""

    def __init__(self, params: Iterable[Tensor], average: bool = True):
        self.active = dist.is_initialized()  # collectives are issued whenever a process group exists (even world 1)
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        self.grads = FlatGradBuffer(params)
        self.average = average

    def view_index(self, step: int, num_views: int) -> int:
        """rank r renders view (step * world + r) mod num_views"""
        return (step * self.world + self.rank) % num_views

    def allreduce_grads(self, async_op: bool = False, prescaled: bool = False):
        """SUM all-reduce of the flat gradient buffer.  ``prescaled``: the gradients were already multiplied by
        1/world at their source (``FusedViewStep(grad_scale=1/world)``), so the averaging pass is skipped."""
        if not self.active:
            return None
        if self.average and self.world > 1 and not prescaled:
            self.grads.flat.div_(self.world)
        return dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def allreduce_densify_stats(self, xyz_gradient_accum: Tensor, denom: Tensor, max_radii2D: Tensor):
        if not self.active:
            return
        packed = torch.stack([xyz_gradient_accum.view(-1).float(), denom.view(-1).float()])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        xyz_gradient_accum.view(-1).copy_(packed[0])
        denom.view(-1).copy_(packed[1])
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)

    def broadcast_params(self, params: Iterable[Tensor], src: int = 0):
        if not self.active:
            return
        for p in params:
            dist.broadcast(p.data, src=src)


class BucketedGradReducer:
    """Gradients in ONE flat fp32 buffer cut into contiguous *buckets* that are all-reduced separately.

    View-parallel schedule (bench.py, world > 1): the backward produces its gradients in two waves -- the SH
    coefficients (60 % of the bytes) are final once the rasterizer backward has run, everything else once the skinning
    backward has run.  Bucket 0 is handed to RCCL while the skinning backward still executes, bucket 1 afterwards, and
    the Adam update of bucket 0 runs while bucket 1 is on the wire.  ``extras[i]`` appends plain float32 scratch to
    bucket i (used for the compact LBS-logit gradient, which replaces the 4x larger dense ``sp_W`` gradient on the wire).

    ``buckets``: list of lists of parameters; every parameter gets ``.grad`` = a view into the flat buffer."""

    def __init__(self, buckets: List[List[Tensor]], extras: Optional[List[int]] = None):
        self.active = dist.is_initialized()
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        extras = list(extras) if extras is not None else [0] * len(buckets)
        assert len(extras) == len(buckets)
        pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731  (every view on a 16-byte boundary, see FlatGradBuffer)
        sizes = [sum(pad4(p.numel()) for p in b) + pad4(e) for b, e in zip(buckets, extras)]
        dev = buckets[0][0].device
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.bucket_views: List[Tensor] = []
        self.extra_views: List[Optional[Tensor]] = []
        off = 0
        for b, e, n in zip(buckets, extras, sizes):
            self.bucket_views.append(self.flat[off:off + n])
            o = off
            for p in b:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
                o += pad4(p.numel())
            self.extra_views.append(self.flat[o:o + e] if e else None)
            off += n

    def view_index(self, step: int, num_views: int) -> int:
        return (step * self.world + self.rank) % num_views

    def allreduce(self, i: int, async_op: bool = True):
        """SUM all-reduce of bucket i (gradients are expected pre-scaled by 1/world at their source)"""
        if not self.active:
            return None
        return dist.all_reduce(self.bucket_views[i], op=dist.ReduceOp.SUM, async_op=async_op)

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


class ShFactorExchange:
    """The SH gradient of a view-parallel step without the 192-byte-per-Gaussian all-reduce.

    The SH gradient of ONE view is rank-1 per Gaussian: basis(view direction) x (clamp-masked colour gradient)
    (gaussian_rasterizer_backwrad.cu:26-127).  ``FusedViewStep(sh_factors=ex.local)`` makes the rasterizer backward write
    those 6 floats per Gaussian instead of the [P,16,3] rows; ``gather()`` all-gathers them in place (24 B per Gaussian
    and rank on the wire) and ``FusedViewStep.sh_grads_from_factors(ex.all, sh_degree)`` rebuilds and sums the rows of every
    view, in rank order, identically on every rank."""

    def __init__(self, P: int, device):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.all = torch.zeros((self.world, P, 6), dtype=torch.float32, device=device)
        self.local = self.all[self.rank]
        self._in_place = self._probe_in_place(device)

    def _probe_in_place(self, device) -> bool:
        """decide ONCE, at construction (outside any timed region), whether the in-place tensor form of the all-gather
        works with this backend / torch build: a small collective is run to completion on every rank and the verdict is
        all-reduced (MIN), so the ranks agree -- an asynchronous RCCL failure inside the training loop cannot be caught by
        an ``except`` around the enqueue, and ranks taking different paths would deadlock"""
        if not dist.is_initialized() or dist.get_backend() != 'nccl':
            return False
        ok = 1
        try:
            buf = torch.zeros((self.world, 8), dtype=torch.float32, device=device)
            buf[self.rank] = float(self.rank + 1)
            dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].view(-1))
            torch.cuda.synchronize(device)
            want = torch.arange(1, self.world + 1, dtype=torch.float32, device=device)[:, None].expand(-1, 8)
            ok = int(torch.equal(buf, want))
        except Exception:  # noqa: BLE001 -- any failure means: use the list form
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    @property
    def nbytes(self) -> int:
        return self.all.numel() * 4

    def gather(self, async_op: bool = False):
        """all-gather the factor blocks; with ``async_op`` returns the work handle (``.wait()`` before the rows are
        rebuilt) so that the exchange can run beside the skinning backward"""
        if not dist.is_initialized():
            return None
        if self._in_place:  # this rank's slice is already where it belongs (form probed at construction)
            return dist.all_gather_into_tensor(self.all.view(-1), self.local.view(-1), async_op=async_op)
        return dist.all_gather(list(self.all.unbind(0)), self.local, async_op=async_op)
