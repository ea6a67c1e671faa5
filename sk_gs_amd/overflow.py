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

The view order of the redone iterations must be reproducible (the loop's own schedule; a seeded sampler).
"""
from typing import Iterable, List, Optional, Tuple

import torch
from torch import Tensor


class OverflowGuard:
    def __init__(self, step, optimizer, every: int = 50, extra_state: Iterable[Tensor] = ()):
        """``step``: the FusedViewStep whose status words are watched; ``optimizer``: FusedAdam (parameters + moments +
        counter are snapshotted); ``extra_state``: further tensors that training mutates (e.g. a sampler's device state)"""
        assert every >= 1
        self.step, self.opt, self.every = step, optimizer, int(every)
        self.extra = list(extra_state)
        self._snap: Optional[List[Tensor]] = None
        self._snap_iter = 0
        self._seen_events = step.status()['overflow_events']
        self.redos = 0
        self.checkpoint(0)  # the state before the first guarded iteration

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
        events = self.step.status()['overflow_events']  # synchronises: once per interval
        if events != self._seen_events:
            self._seen_events = events
            torch._foreach_copy_(self._live_tensors(), self._snap)
            self.redos += 1
            return 'redo', self._snap_iter
        self.checkpoint(iteration + 1)
        return None

    def rebind(self, step):
        """after the caller rebuilt its FusedViewStep (bigger capacity): watch the new one"""
        self.step = step
        self._seen_events = step.status()['overflow_events']
