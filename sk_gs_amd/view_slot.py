"""One captured graph for every training view: the per-view inputs of the step as a device-resident "view slot".

The reference builds the rasterizer settings of each view on the host (``prepare_inputs``, networks/gaussian_splatting.py:
271-284) and walks its views in the training loop (train.py:179-250, datasets/DNerfDataset.py:231-261).  Capturing the
step as a hipGraph bakes every host-side scalar and pointer into the graph, so round 1 kept one graph per view: capture
time and graph memory grew with the number of views (D-NeRF: 100-200, ZJU-MoCap: thousands) and every densification
re-captured all of them.

Here everything that changes from view to view is a DEVICE LOAD of the kernels:

    word  0..15  viewmatrix      (skgs_raster_inputs.viewmatrix points here)
    word 16..31  projmatrix      (.projmatrix)
    word 32..34  campos          (.campos)
    word 36..37  tanfovx, tanfovy  (.tanfov_device)
    word 38      time t of the frame (float; the deform network's ``t`` argument points here)
    word 39      frame index (int32; skgs_bone_chain_*'s ``frame_index``: row of the global_tr table)
    word 40      target index (int32; skgs_image_loss_*'s ``gt_index``: image of the target stack)

``ViewTable`` keeps one such 256-byte record per view on the device; ``select(v)`` copies record v into the live slot
with ONE small device-to-device copy on the current stream (no host synchronisation), after which the SAME captured
graph renders / trains view v.  Image size, SH degree, ``colmap`` and the other settings are common to all views of a
table (they stay launch arguments).
"""
import ctypes as C
from typing import Optional, Sequence

import torch
from torch import Tensor

from sk_gs_amd.renderer.gaussian_render import GaussianRasterizationSettings

SLOT_WORDS = 64
W_VIEW, W_PROJ, W_CAMPOS, W_TANFOV, W_TIME, W_FRAME, W_TARGET = 0, 16, 32, 36, 38, 39, 40


def view_record(rs: GaussianRasterizationSettings, time: float, frame_index: int, target_index: int) -> Tensor:
    """the 64-word record of one view (CPU float32 tensor; the two indices are stored as int32 bit patterns)"""
    rec = torch.zeros(SLOT_WORDS, dtype=torch.float32)
    rec[W_VIEW:W_VIEW + 16] = rs.viewmatrix.detach().float().cpu().reshape(16)
    rec[W_PROJ:W_PROJ + 16] = rs.projmatrix.detach().float().cpu().reshape(16)
    rec[W_CAMPOS:W_CAMPOS + 3] = rs.campos.detach().float().cpu().reshape(3)
    rec[W_TANFOV], rec[W_TANFOV + 1] = float(rs.tanfovx), float(rs.tanfovy)
    rec[W_TIME] = float(time)
    rec.view(torch.int32)[W_FRAME] = int(frame_index)
    rec.view(torch.int32)[W_TARGET] = int(target_index)
    return rec


class ViewAdvance(C.Structure):
    """include/skgs.h::skgs_view_advance: the next view's record is put into the slot by the launch that closes a step"""
    _fields_ = [('table', C.c_void_p), ('order', C.c_void_p), ('cursor', C.c_void_p), ('slot', C.c_void_p),
                ('n_order', C.c_int32), ('words', C.c_int32)]


class ViewTable:
    """records of all training views + the stack of their target images, and the live slot the kernels read"""

    def __init__(self, settings: Sequence[GaussianRasterizationSettings], times: Sequence[float],
                 frame_indices: Sequence[int], targets: Optional[Tensor], device, target_indices: Optional[Sequence[int]] = None):
        assert len(settings) == len(times) == len(frame_indices) > 0
        first = settings[0]
        for rs in settings:  # everything that stays a launch argument must agree
            assert (rs.image_height, rs.image_width, rs.sh_degree, bool(rs.colmap), float(rs.scale_modifier),
                    bool(rs.prefiltered)) == (first.image_height, first.image_width, first.sh_degree, bool(first.colmap),
                                              float(first.scale_modifier), bool(first.prefiltered)), \
                'all views of a ViewTable share the image size, SH degree, colmap, scale_modifier and prefiltered'
        self.settings = first
        self.n_views = len(settings)
        if target_indices is None:
            target_indices = list(range(self.n_views))
        if targets is not None:
            assert targets.dim() == 4 and targets.is_contiguous() and targets.dtype == torch.float32
            assert max(target_indices) < targets.shape[0]
            assert tuple(targets.shape[2:]) == (first.image_height, first.image_width)
            targets = targets.to(device)
        self.targets = targets
        recs = torch.stack([view_record(rs, t, f, ti) for rs, t, f, ti in zip(settings, times, frame_indices, target_indices)])
        self.records = recs.to(device)
        self.slot = self.records[0].clone()
        self.current = 0

    def select(self, v: int):
        """make view ``v`` the one the next launches (or graph replays) see: one 256-byte device-to-device copy"""
        self.slot.copy_(self.records[v], non_blocking=True)
        self.current = v

    def set_order(self, order: Sequence[int], capacity: Optional[int] = None):
        """Walk the views in ``order`` (cyclically) WITHOUT a per-step ``select``: view ``order[0]`` is selected now; the
        launch that closes a step (``FusedAdam.step_tail(next_view=table.advance())``, what ``FusedTrainStep`` issues) then
        puts ``order[1]``, ``order[2]``, ... into the slot, one per step -- the device-to-device copy of ``select`` (4.5 us
        in front of every replay) disappears.  Upload a new order (a new epoch's permutation) at any step boundary.

        The order, its length and the cursor live in storage allocated ONCE (the first call; ``capacity`` entries, default
        4 x the number of views): captured graphs hold these addresses as kernel arguments, so every later call copies IN
        PLACE.  An order longer than the capacity cannot be served by graphs captured before it: that raises."""
        assert len(order) >= 1 and all(0 <= int(v) < self.n_views for v in order)
        dev = self.records.device
        if getattr(self, '_order_store', None) is None:
            cap = max(int(capacity or 0), 4 * self.n_views, len(order))
            self._order_store = torch.zeros(cap, dtype=torch.int32, device=dev)
            self.cursor = torch.zeros(2, dtype=torch.int32, device=dev)  # {next position, length of the order}
        if len(order) > self._order_store.numel():
            raise ValueError(f'an order of {len(order)} views does not fit the {self._order_store.numel()} entries allocated by '
                             f'the first set_order (captured graphs point at that storage): pass capacity= to the first call')
        host = torch.tensor([int(v) for v in order], dtype=torch.int32)
        self._order_store[:len(order)].copy_(host, non_blocking=False)
        self.order = self._order_store[:len(order)]  # a view: same storage, the current length
        self.cursor.copy_(torch.tensor([1, len(order)], dtype=torch.int32))  # the next view to load is order[1]
        self._order_host = [int(v) for v in order]
        self.select(int(order[0]))

    def seek(self, i: int):
        """position the walk so that the next step trains view ``order[i % len(order)]`` (roll-back of an OverflowGuard:
        redo from iteration i of a loop that follows the order), in place"""
        assert getattr(self, 'order', None) is not None
        n = len(self._order_host)
        self.cursor[0:1].fill_(int(i) + 1)
        self.select(self._order_host[int(i) % n])

    def rewind(self):
        """back to ``order[0]``, in place (captured graphs keep pointing at the same order / cursor storage)"""
        self.seek(0)

    def clear_order(self):
        """back to explicit ``select`` calls (graphs captured with the self-advancing closing launch keep advancing the
        cursor: capture again without ``next_view`` to stop that)"""
        self.order = None

    def state_tensors(self):
        """the device tensors a training loop mutates through this table (cursor + live slot): hand them to
        ``OverflowGuard(extra_state=...)`` so that a roll-back also restores the position in the view order"""
        return [t for t in (getattr(self, 'cursor', None), self.slot) if t is not None]

    def advance(self) -> Optional[ViewAdvance]:
        """the ``skgs_view_advance`` job of the closing launch (None until ``set_order`` was called)"""
        if getattr(self, 'order', None) is None:
            return None
        return ViewAdvance(self.records.data_ptr(), self._order_store.data_ptr(), self.cursor.data_ptr(), self.slot.data_ptr(),
                           0, SLOT_WORDS)  # n_order = 0: the length is the device word cursor[1]

    # ---- device addresses of the live slot's fields
    def ptr(self, word: int) -> int:
        return self.slot.data_ptr() + 4 * word

    def settings_of(self, v: int) -> dict:
        """the host-side view of record v (tests)"""
        r = self.records[v].cpu()
        return dict(viewmatrix=r[W_VIEW:W_VIEW + 16].view(4, 4), projmatrix=r[W_PROJ:W_PROJ + 16].view(4, 4),
                    campos=r[W_CAMPOS:W_CAMPOS + 3], tanfovx=float(r[W_TANFOV]), tanfovy=float(r[W_TANFOV + 1]),
                    time=float(r[W_TIME]), frame_index=int(r.view(torch.int32)[W_FRAME]),
                    target_index=int(r.view(torch.int32)[W_TARGET]))
