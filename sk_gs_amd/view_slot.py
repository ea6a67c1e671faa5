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

    def set_order(self, order: Sequence[int]):
        """Walk the views in ``order`` (cyclically) WITHOUT a per-step ``select``: view ``order[0]`` is selected now; the
        launch that closes a step (``FusedAdam.step_tail(next_view=table.advance())``, what ``FusedTrainStep`` issues) then
        puts ``order[1]``, ``order[2]``, ... into the slot, one per step -- the device-to-device copy of ``select`` (4.5 us
        in front of every replay) disappears.  Upload a new order (a new epoch's permutation) at any step boundary."""
        assert len(order) >= 1 and all(0 <= int(v) < self.n_views for v in order)
        dev = self.records.device
        self.order = torch.tensor([int(v) for v in order], dtype=torch.int32, device=dev)
        self.cursor = torch.ones(1, dtype=torch.int32, device=dev)  # the next view to load is order[1]
        self.select(int(order[0]))

    def rewind(self):
        """back to ``order[0]``, in place (captured graphs keep pointing at the same order / cursor storage)"""
        assert getattr(self, 'order', None) is not None
        self.cursor.fill_(1)
        self.select(int(self.order[0].item()))

    def clear_order(self):
        """back to explicit ``select`` calls"""
        self.order = None

    def advance(self) -> Optional[ViewAdvance]:
        """the ``skgs_view_advance`` job of the closing launch (None until ``set_order`` was called)"""
        if getattr(self, 'order', None) is None:
            return None
        return ViewAdvance(self.records.data_ptr(), self.order.data_ptr(), self.cursor.data_ptr(), self.slot.data_ptr(),
                           int(self.order.numel()), SLOT_WORDS)

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
