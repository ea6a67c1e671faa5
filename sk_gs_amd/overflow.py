"""Recovery from a binning-capacity overflow in a sync-free training loop.

The hipGraph-replayed step never asks the host how many tile instances a view produced (the reference blocks on that
number in every forward, gaussian_rasterizer_forward.cu:209): the binning buffer has a fixed capacity (or fixed per-tile
buckets) and a forward that needs more DROPS the excess and bumps a sticky device counter (``skgs_status.overflow_events``).
Training moves and grows the Gaussians, so a capacity that fitted at capture time can stop fitting -- and a step with
dropped splats has a truncated render and truncated gradients.  Nothing may be built on such a step.

``OverflowGuard`` makes that safe without a per-step synchronisation:

  * every ``every`` steps it reads the counter (one small synchronising read per interval) and, if no forward overflowed
    since the last checkpoint, snapshots the training state -- parameters, Adam moments and step counter, densification
    statistics: device-to-device copies, ~100 MB / 40 us at config #1, amortised over the interval;
  * if a forward DID overflow it restores the last snapshot and tells the caller from which iteration to redo; the caller
    grows the capacity (``FusedViewStep.grow_capacity``), re-captures its graphs and replays those iterations.  The redone
    steps see complete tile lists, so the run continues as if the capacity had always been large enough.

The view order of the redone iterations must be reproducible (the loop's own schedule; a seeded sampler).  A loop that lets
the closing launch walk an ordered ``ViewTable`` hands the table to the guard (``view_table=``): its cursor and live slot are
then part of the snapshot, so the redone iterations see the same views.  The snapshot is tied to the number of Gaussians:
call ``checkpoint()`` after every densification / pruning.  A fused deform network whose in-launch exchange timed out
(``status()['mlp_failed']``) left partial gradients behind: the guard treats that like an overflow (roll back, redo).
"""
from typing import Iterable, List, Optional, Tuple

import torch
from torch import Tensor


class OverflowGuard:
    def __init__(self, step, optimizer, every: int = 50, extra_state: Iterable[Tensor] = (), view_table=None):
        """``step``: the FusedViewStep whose status words are watched; ``optimizer``: FusedAdam (parameters + moments +
        counter are snapshotted); ``extra_state``: further tensors that training mutates (e.g. a sampler's device state)"""
        assert every >= 1
        self.step, self.opt, self.every = step, optimizer, int(every)
        self.extra = list(extra_state) + (view_table.state_tensors() if view_table is not None else [])
        self._snap: Optional[List[Tensor]] = None
        self._snap_iter = 0
        self._seen_events = self._events(step)
        self.redos = 0
        self.checkpoint(0)  # the state before the first guarded iteration

    @staticmethod
    def _events(step) -> Tuple[int, int]:
        st = step.status()
        return st['overflow_events'], st.get('mlp_failed', 0)

    def _live_tensors(self) -> List[Tensor]:
        ts = []
        for p in self.opt.params:
            ts += [p.data, self.opt.state[p]['exp_avg'], self.opt.state[p]['exp_avg_sq']]
        ts.append(self.opt.step_state)
        ts += [self.step.xyz_gradient_accum, self.step.denom, self.step.max_radii2D]
        return ts + self.extra

    @torch.no_grad()
    def checkpoint(self, iteration: int):
        """snapshot the state as it is BEFORE iteration ``iteration`` runs"""
        live = self._live_tensors()
        if self._snap is None or len(self._snap) != len(live) or any(a.shape != b.shape for a, b in zip(self._snap, live)):
            self._snap = [t.clone() for t in live]
        else:
            torch._foreach_copy_(self._snap, live)
        self._snap_iter = iteration

    @torch.no_grad()
    def after_step(self, iteration: int) -> Optional[Tuple[str, int]]:
        """call after iteration ``iteration`` has been issued.  Returns None, or ('redo', first_iteration): the state has
        been rolled back to the snapshot taken before ``first_iteration``; grow the capacity, re-capture, replay from it
        (a ``FusedTrainStep(pre_forward=True)`` carries the NEXT view's skeleton state across steps: ``prime()`` it again
        after the roll-back, once the view to resume from is in the slot)."""
        if (iteration + 1) % self.every:
            return None
        return self.check_now(iteration)

    @torch.no_grad()
    def check_now(self, iteration: int) -> Optional[Tuple[str, int]]:
        """the check of ``after_step`` regardless of the interval (e.g. right before a densification, which must not be
        decided on statistics a truncated step contributed to)"""
        events = self._events(self.step)  # synchronises: once per interval
        if events != self._seen_events:
            self._seen_events = events
            live = self._live_tensors()
            if len(live) != len(self._snap) or any(a.shape != b.shape for a, b in zip(live, self._snap)):
                raise RuntimeError('OverflowGuard: the training state changed shape since the last snapshot (densification / '
                                   'pruning?): call checkpoint(iteration) right after such an operation')
            torch._foreach_copy_(live, self._snap)
            self.redos += 1
            return 'redo', self._snap_iter
        self.checkpoint(iteration + 1)
        return None

    def rebind(self, step):
        """after the caller rebuilt its FusedViewStep (bigger capacity): watch the new one"""
        self.step = step
        self._seen_events = self._events(step)
