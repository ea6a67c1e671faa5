"""hipGraph capture of the training step, and the one-rank step with its optimizer folded into the backward.

``GraphedSteps``: the step has static shapes and, with the capacity-based binning (``_C.config.sync_num_rendered =
False``) or the bucket layout, no host synchronisation: it replays as ONE hipGraph -- for all training views when the
per-view inputs live in a ``ViewTable`` slot.  (The reference cannot do this: its forward blocks on a D2H copy of
``num_rendered``, gaussian_rasterizer_forward.cu:209.)  Through the autograd operator path a step is ~15 of our kernels plus
the torch glue around them; ``FusedViewStep`` issues 14 launches and no glue.

``FusedTrainStep``: ``FusedViewStep.forward_backward`` + ``FusedAdam.step`` for one rank, with the update taken apart along
the data dependencies of the skeleton stage's backward launch.
"""
import gc
from typing import Callable, Dict, Hashable, Optional

import torch


class GraphedSteps:
    """Captures ``fn(key)`` once per key into a ``torch.cuda.CUDAGraph`` and replays it afterwards.

    ``fn`` must be free of host synchronisation and must only touch tensors that live across calls (parameters,
    optimizer state, static inputs) or that it allocates itself (those come from the graph's private pool).

    Pitfall (found the hard way on ROCm 7.2): no autograd graph built on ANOTHER stream may still be alive when a
    capture starts.  A parameter's AccumulateGrad node is cached while any graph references it and remembers the
    stream it was created on; a backward captured on the side stream would then synchronise with that foreign
    (legacy default) stream, which drags it into the capture and crashes hipStreamEndCapture.  Drop such outputs
    (or run those forwards under ``torch.no_grad()``) before capturing; ``capture`` runs ``gc.collect()`` first.

    ``capture`` EXECUTES ``fn`` ``warmup`` times (default: once) before recording it: lazily initialised library state
    must exist before a capture starts.  View-parallel training must therefore capture an optimizer step only where it
    would run anyway -- after the gradient all-reduce -- or the ranks' replicas diverge.  Calling the object with a key
    that has no graph yet captures it and does NOT replay on top: the warm-up execution already was that call's one
    execution (an optimizer step or a statistics update must not run two or three times on the first call)."""

    def __init__(self, fn: Callable[[Hashable], None], warmup: int = 1, collect_garbage: bool = True, thread_local: bool = False):
        """``collect_garbage``: run ``gc.collect()`` before a capture (see the pitfall above).  A step that builds no
        autograd graph (``FusedViewStep``) does not need it; the collection is most of the cost of re-capturing after a
        densification (~20 of ~27 ms)."""
        self.fn = fn
        self.warmup = warmup
        self.collect_garbage = collect_garbage
        # capture_error_mode 'thread_local': other threads of the process may touch the device while a capture is open (a data
        # loader's pin-memory thread under the reference's training loop: sk_gs_amd/reference_fused.py)
        self.thread_local = bool(thread_local)
        self.graphs: Dict[Hashable, torch.cuda.CUDAGraph] = {}
        self.pool = None

    def capture(self, key: Hashable, repeat: int = 1, warmup: Optional[int] = None):
        """``repeat`` > 1: ``fn(key)`` recorded ``repeat`` times in ONE graph (stored beside the one-step graph of the key) --
        for a step whose state lives on the device (an ordered ``ViewTable``: the closing launch selects the next view) this is
        ``repeat`` consecutive training steps per replay.  Between two replays on a stream the device idles for ~8 us
        (rocprofv3: tools/step_timeline.py); four steps per graph took 4 us off each 0.35 ms step of bench.py.  The warm-up
        executions before the recording run ``fn`` once each, whatever ``repeat`` is; ``warmup=0`` skips them (the one-step
        graph of the key exists: everything lazy has been initialised, and no extra training step is executed)."""
        if self.collect_garbage:
            gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup if warmup is None else warmup):
                self.fn(key)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # With a process group alive its watchdog thread polls the events of earlier (eager) collectives every 100 ms:
        # hipEventQuery from ANOTHER thread is an error while a capture in the default "global" mode is open, the watchdog
        # rethrows it and the rank aborts (seen once in ~100 one-rank RCCL runs of bench.py).  "thread_local" restricts the
        # check to this thread, which is the one that captures.
        import torch.distributed as dist
        mode = 'thread_local' if (self.thread_local or (dist.is_available() and dist.is_initialized())) else 'global'
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):
            for _ in range(repeat):
                self.fn(key)
        if self.pool is None:
            self.pool = g.pool()
        self.graphs[key if repeat == 1 else (key, '*', repeat)] = g
        return g

    def replay(self, key: Hashable, repeat: int):
        """replay the ``repeat``-step graph of ``key`` (captured with ``capture(key, repeat)``)"""
        self.graphs[(key, '*', repeat)].replay()

    def __call__(self, key: Hashable):
        g = self.graphs.get(key)
        if g is None:
            if self.warmup >= 1:
                if self.warmup > 1:  # exactly one real execution on this call
                    saved, self.warmup = self.warmup, 1
                    try:
                        self.capture(key)
                    finally:
                        self.warmup = saved
                else:
                    self.capture(key)
                return
            g = self.capture(key)
        g.replay()


class FusedTrainStep:
    """One training step of one rank: ``FusedViewStep.forward_backward`` + ``FusedAdam.step`` with the optimizer taken
    apart along the data dependencies of the skeleton stage's backward launch (``skgs_skeleton_backward``), a dependent chain
    on 32 workgroups (~45 us with the kinematic chain) that leaves 224 CUs idle:

      * the per-Gaussian parameters (xyz, SH, opacity, scaling, rotation, LBS logits: 95 % of the optimizer's bytes) have
        final gradients before it starts and are not touched by it: workgroups 32.. of the SAME launch stream their update;
      * network, joint positions and per-frame tables follow in one short launch that advances the step counter itself
        (last workgroup out) and whose workgroup for the joints first completes their gradient (the frequency-encoding
        backward of the network-input path) -- no launch between backward and update.

    As separate launches (Adam, counter, encoder backward) the tail of the step was 50 + 38 + 4 + 5 us; it is 57 + 8.
    (Updating the network's weights inside the backward launch as well, by the workgroups that own their rows, was built
    and measured: the launch grows by 15 us, more than the short launch costs.)  Same arithmetic as
    ``step.forward_backward(); optimizer.step()`` (tests/test_gpu_optim.py: bit-identical updates).  Falls back to exactly
    that when the step has no fused network (or ``enable=False``).  With a ``ViewTable`` whose order was set
    (``set_order``) the closing launch also selects the next view: no per-step ``select``.

    ``pre_forward`` (needs that ordered table): the step ENDS with the skeleton-forward launch of the NEXT view -- the slot
    already names it, network and joints already carry this step's update -- and the rows' update is split between the
    two 32-workgroup launches, ``ROWS_IN_BACKWARD`` of the chunks beside the skeleton backward, the rest beside that forward
    (after the counter moved: ``skgs_adam_range.after_advance``).  Every parameter is final when the step returns; the next
    step starts at the skinning.  Call ``prime()`` once before the first step (and after anything that changes network,
    joints or the slot from outside: a restore, a ``select``).  Measured (tools/time_skeleton.py, the two launches alone):
    100k Gaussians 90 -> 90 us (the whole update already hides beside the backward, and the stream slows the network's
    hand-offs wherever it runs), 200k 125 -> 100 on one box and 112 -> 114 on another, 500k 251 -> 227 / 233 -> 204.
    ``pre_forward='auto'`` therefore switches it on only when the rows' update moves more than ``AUTO_BYTES`` (BASELINE
    configs #3 and #4: +2.8 % and +1..2 % of the step).

    ``reduce_between`` (view-parallel ranks): a gradient all-reduce runs between the backward and the update, so nothing of
    the optimizer can ride on the backward launch.  The step is then two calls, ``backward()`` -- all-reduce -- ``update()``;
    with ``pre_forward`` the update is the closing launch (network, joints, tables, counter, next view) followed by the next
    view's skeleton-forward launch with ALL rows as its side job: two launches (9 + ~50 us at 100k Gaussians) where a full Adam
    launch and, in the next step, a bare skeleton forward were (41 + 32 us)."""

    ROW_GROUPS = ('xyz', 'f_dc', 'f_rest', 'opacity', 'scaling', 'rotation', 'sp_W')
    ROWS_IN_BACKWARD = 0.6   # ~ backward launch time / (backward + forward launch time) of the bare skeleton stage
    AUTO_BYTES = 600e6       # 28 B per element: ~270k Gaussians with degree-3 SH and 20 bones

    def __init__(self, step, optimizer, enable: bool = True, pre_forward: bool = False, reduce_between: bool = False):
        self.step, self.optimizer = step, optimizer
        self.reduce_between = bool(reduce_between)
        names = [g.get('name') for g in optimizer.param_groups]
        self.rows = [n for n in names if n in self.ROW_GROUPS]
        self.rest = [n for n in names if n not in self.ROW_GROUPS]
        # (rows rebuilt after the backward -- compact logit gradient, SH factors -- are final only when update() starts)
        # (distance-based LBS weightings: their backward re-reads xyz AFTER the skeleton backward launch, which must therefore
        # not carry the rows' update -- the plain sequence forward_backward(); optimizer.step() serves them)
        self.fused = bool(enable and self.rows and self.rest and getattr(step, '_mlp_fused', None) is not None
                          and getattr(step, 'lbs_method', 'W') == 'W'
                          and (reduce_between or (step.spw_logit_grad is None and step.sh_factors is None))
                          and len(optimizer._chunk_ranges(self.rows)) == 1 and len(optimizer._chunk_ranges(self.rest)) == 1)
        self.set_pre_forward(pre_forward)
        # the joints' gradient through the network input is completed by the optimizer's closing launch (one rank; with an
        # all-reduce in between the backward completes it itself: the exchange needs the whole gradient)
        self.joints = step.model.joints if (self.fused and not reduce_between and step.input_grad_job() is not None) else None
        step.defer_input_grad = self.joints is not None

    def set_pre_forward(self, on: bool):
        """switch the mode (off: before anything else uses the step's ``forward`` or selects views by hand)"""
        step = self.step
        if on == 'auto':
            on = self.fused and 28 * sum(p.numel() for g in self.optimizer.param_groups if g.get('name') in self.rows
                                         for p in g['params']) > self.AUTO_BYTES
        self.pre_forward = bool(on and self.fused and step.view_table is not None)
        part = (0.0, self.ROWS_IN_BACKWARD) if self.pre_forward else None
        step.side_optimizer = (self.optimizer, self.rows, part) if (self.fused and not self.reduce_between) else None
        step.skeleton_ahead = self.pre_forward

    def backward(self, rs=None, time_id=None, target=None):
        """``reduce_between``: forward + backward of the view (the gradients then go through the all-reduce)"""
        self.step.forward_backward(rs, time_id, target)

    def update(self):
        """``reduce_between``: the optimizer step on the reduced gradients"""
        assert self.reduce_between
        if not self.pre_forward:
            self.optimizer.step()
            return
        vt = self.step.view_table
        assert vt.advance() is not None, 'pre_forward steps take their views from the ordered table'
        self.optimizer.step_tail(self.rest, next_view=vt.advance())
        self.step.skeleton_forward(None, side_adam=self.optimizer.side_range(self.rows, after_advance=True))

    def __call__(self, rs=None, time_id=None, target=None):
        assert not self.reduce_between, 'a step with an all-reduce in it is backward() ... update()'
        self.step.forward_backward(rs, time_id, target)
        self._tail(rs)

    def loss(self, rs=None, time_id=None, target=None):
        """The same step behind torch's autograd API (the reference's loop shape, train.py:179-250)::

            loss = train.loss(rs, time_id, target)     # the forward half runs now
            loss.backward()                            # the backward half; the per-Gaussian rows' Adam update rides on its skeleton launch
            optimizer.step()                           # = the closing launch (network, joints, tables, counter, next view)

        Same launches and the same arithmetic as ``train(rs, time_id, target)``.  What differs from a plain optimizer: the rows are
        already updated when ``backward()`` returns (their gradients are final before the skeleton launch starts), so nothing may
        read or change those gradients between ``backward()`` and ``optimizer.step()`` -- the next ``optimizer.step()`` call is
        consumed by this step's tail."""
        assert not self.reduce_between, 'a step with an all-reduce in it is backward() ... update()'
        if self.optimizer._pending_tail is not None:
            # the previous view's backward ran (its skeleton launch carried the rows' update) but optimizer.step() -- that step's
            # closing launch -- was never called: network / joints / counter are one update behind the rows
            raise RuntimeError('FusedTrainStep.loss(): the previous loss.backward() was not followed by optimizer.step()')
        # the tail is armed by the node's BACKWARD, after the launches that carry the rows' update (armed in the forward, a
        # loss evaluated without backward() -- logging, torch.no_grad(), an exception, a skipped NaN step -- turned a later, unrelated
        # optimizer.step() into a closing launch on stale gradients)
        self.step._arm_tail = (self.optimizer, (self, rs))
        return self.step.loss(rs, time_id, target)

    def _tail(self, rs):
        if self.fused:
            job = self.step.input_grad_job() if self.joints is not None else None
            vt  = self.step.view_table
            self.optimizer.step_tail(self.rest, freq_job=job, freq_param=self.joints,
                                     next_view=vt.advance() if (vt is not None and rs is None) else None)
            if self.pre_forward:
                assert rs is None and vt.advance() is not None, 'pre_forward steps take their views from the ordered table'
                self.step.skeleton_forward(None, side_adam=self.optimizer.side_range(
                    self.rows, (self.ROWS_IN_BACKWARD, 1.0), after_advance=True))
        else:
            self.optimizer.step()

    def prime(self):
        """pre_forward: the skeleton stage of the view in the slot, before the first step"""
        if self.pre_forward:
            self.step.skeleton_forward(None)
