"""Fused multi-tensor Adam (scope row (f)-2): every parameter of every group is updated by ONE kernel launch
(csrc/adam.hip) instead of torch's per-group ``multi_tensor_apply`` launches.

Semantics = ``torch.optim.Adam(params, betas, eps, amsgrad=False, weight_decay=0)`` as the reference configures it
(exps/default.yaml:122-125: eps 1e-15, betas (0.9, 0.999); per-group learning rates networks/gaussian_splatting.py:
443-453).  The step counter and the descriptor table live on the device, so ``step()`` is hipGraph-capturable;
learning-rate changes (``update_learning_rate``, gaussian_splatting.py:455-465) are pushed with ``set_lr`` outside
the graph.
"""
import ctypes as C
import struct
from typing import Iterable, List

import torch

from sk_gs_amd import _C
from sk_gs_amd.capacity import cap_store, regrad, slot_numel


def position_lr(step: int, lr_init: float, lr_final: float, max_steps: int = 30_000, delay_steps: int = 0,
                delay_mult: float = 1.0) -> float:
    """Learning rate of the `xyz` group at ``step``: log-linear interpolation from ``lr_init`` to ``lr_final`` over
    ``max_steps``, optionally eased in by a sine ramp from ``delay_mult`` over ``delay_steps`` -- the schedule
    ``get_expon_lr_func`` builds for ``update_learning_rate`` (networks/gaussian_splatting.py:56-84,455-465; defaults
    lr_position_init 0.16e-3 -> lr_position_final 0.0016e-3 over 30k steps).  Push it with ``FusedAdam.set_lr``."""
    import math
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    ramp = 1.0
    if delay_steps > 0:
        ramp = delay_mult + (1.0 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return ramp * math.exp((1.0 - t) * math.log(lr_init) + t * math.log(lr_final))


class AdamRange(C.Structure):
    """include/skgs.h::skgs_adam_range: a run of chunks of an optimizer table, updated as the side job of another launch"""
    _fields_ = [('n_tensors', C.c_int32), ('tensors', C.c_void_p), ('chunk_begin', C.c_int64), ('chunk_end', C.c_int64),
                ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double), ('step_count', C.c_void_p),
                ('after_advance', C.c_int32)]


class CapacityExceeded(RuntimeError):
    """an in-place densification needs more rows than the row capacity holds: re-home the model (enable_capacity with a
    larger P_cap), rebuild gradient buffers / optimizer state / step and re-capture"""


class FusedAdam:
    MAX_CAPTURED_TABLES = 64  # captured steps with gradients of their own (one per graph: e.g. one graph per training view)

    def __init__(self, param_groups: Iterable[dict], betas=(0.9, 0.999), eps: float = 1e-15,
                 zero_after_step: 'torch.Tensor' = None):
        lib = _C.load_library()
        lib.skgs_adam_chunk_elems.restype = C.c_int64
        lib.skgs_adam_tensor_bytes.restype = C.c_size_t
        self.param_groups: List[dict] = [dict(g) for g in param_groups]
        self.betas, self.eps = betas, eps
        # contiguous float32 gradient storage cleared by the step itself (see FusedViewStep(tables_zeroed_by_optimizer))
        self.zero_after_step = zero_after_step
        assert zero_after_step is None or (zero_after_step.is_contiguous() and zero_after_step.dtype == torch.float32)
        self._chunk = int(lib.skgs_adam_chunk_elems())
        assert int(lib.skgs_adam_tensor_bytes()) == 56
        self._state_listeners = []  # callables run after the moments changed from outside a step (see add_state_listener)
        self._schedules, self._sched_of_group, self._sched_dev = [], {}, None  # device learning-rate schedules (set_lr_schedule)
        self.params, self._lr_index = [], []
        for gi, g in enumerate(self.param_groups):
            g['params'] = [p for p in g['params']]
            for p in g['params']:
                if p.requires_grad:
                    assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                    self.params.append(p)
                    self._lr_index.append(gi)
        dev = self.params[0].device
        # (a per-Gaussian parameter with a row capacity -- sk_gs_amd/capacity.py -- gets moments of the same capacity)
        self._cap_state = {}
        self.state = {}
        for p in self.params:
            store = cap_store(p)
            if store is None:
                self.state[p] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
            else:
                m, v = torch.zeros_like(store), torch.zeros_like(store)
                self._cap_state[p] = (m, v)
                self.state[p] = dict(exp_avg=m[:p.shape[0]], exp_avg_sq=v[:p.shape[0]])
        # the optimizer's device state (include/skgs.h::skgs_adam_step): word 0 = steps taken (float), doubles at bytes 8 /
        # 16 = 1 - beta^steps, maintained by the kernels; all zeros = no step taken
        lib.skgs_adam_state_bytes.restype = C.c_size_t
        self.step_state = torch.zeros(int(lib.skgs_adam_state_bytes()) // 4, dtype=torch.float32, device=dev)
        self._pending_tail = None  # (FusedTrainStep.loss: the next step() is that training step's closing launch)
        self.step_count = self.step_state[:1]  # (a view: ``float(opt.step_count)`` reads the count)
        self._table = torch.zeros(len(self.params) * 56, dtype=torch.uint8, device=dev)
        # descriptor tables of captured steps whose gradients autograd handed over (_table_of_this_capture)
        self._capture_tables = []
        self._retired_arenas = []
        self._new_capture_arenas()
        self._total_chunks = 0
        self._bound_grads = None
        self._upload()

    # ---------------------------------------------------------------------------------------------------------
    def _upload(self):
        """(re)build the device descriptor table; must run outside graph capture (H2D copy)"""
        blob, chunk0 = bytearray(), 0
        grads = []
        self._chunk0 = []  # first chunk of every parameter (+ the total at the end): ranges for partial steps
        for p, gi in zip(self.params, self._lr_index):
            self._chunk0.append(chunk0)
            if p.grad is None:
                store = cap_store(p)
                if store is None:
                    p.grad = torch.zeros_like(p)
                else:
                    p.grad, p._grad_slot = torch.zeros_like(store)[:p.shape[0]], store.numel()
            st = self.state[p]
            n = p.numel()
            blob += struct.pack('<QQQQqqfi', p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(),
                                st['exp_avg_sq'].data_ptr(), n, chunk0, float(self.param_groups[gi]['lr']), self._sched_slot(gi))
            # chunk space by CAPACITY: ranges handed to captured launches stay valid when the live row count changes (the
            # kernels read n from this table and skip the rest of a slot)
            chunk0 += (slot_numel(p) + self._chunk - 1) // self._chunk
            grads.append(p.grad.data_ptr())
        self._chunk0.append(chunk0)
        self._total_chunks = chunk0
        self._bound_grads = grads
        self._h2d(self._table, blob)

    def _new_capture_arenas(self, slots: int = None):
        """(re)create the arenas the per-capture descriptor tables are carved from: a pinned host blob and a device table per
        slot, sized for the CURRENT number of parameters.  Outside graph capture only (pinned memory cannot be allocated while
        a stream captures).  The previous arenas are kept alive -- graphs captured earlier may still replay copy nodes that
        read them -- until ``release_captured_tables(drop_retired=True)``."""
        slots = max(int(slots or 0), self.MAX_CAPTURED_TABLES)
        if getattr(self, '_pin_arena', None) is not None and self._capture_tables:
            self._retired_arenas.append((self._pin_arena, self._table_arena))
        n = slots * len(self.params) * 56
        self._capture_slots = slots
        self._pin_arena = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        self._table_arena = torch.zeros(n, dtype=torch.uint8, device=self.params[0].device)
        self._capture_tables = []

    def release_captured_tables(self, drop_retired: bool = True, slots: int = None):
        """Forget the descriptor tables of captured steps (``_table_of_this_capture``): call it when the graphs that own them
        are discarded -- e.g. before re-capturing after a densification -- so their slots are used again.  The optimizer does
        this by itself whenever parameter or moment ADDRESSES change (``change_optimizer`` / ``gather_rows`` without a row
        capacity: every graph captured before is invalid then anyway).  ``slots``: reserve room for that many captured steps
        (default ``MAX_CAPTURED_TABLES``); ``drop_retired``: also free arenas retired earlier (no old graph replays again)."""
        assert not torch.cuda.is_current_stream_capturing(), 'release_captured_tables: outside graph capture'
        self._capture_tables = []
        if drop_retired:
            self._retired_arenas = []
        target = max(int(slots or 0), self.MAX_CAPTURED_TABLES)
        if target != self._capture_slots or self._pin_arena.numel() != target * len(self.params) * 56:
            if not drop_retired:
                self._retired_arenas.append((self._pin_arena, self._table_arena))
            self._new_capture_arenas(target)

    def _addresses_changed(self):
        """parameter / moment tensors were re-created (densification without a row capacity): every graph captured before is
        invalid, so are the descriptor tables made for them -- their slots are free again; the arenas are re-made when the
        number of parameters changed (a slot is n_params x 56 bytes).  The old arenas stay alive until released: a copy node
        of a stale graph must never read freed pinned memory."""
        if self._pin_arena.numel() != self._capture_slots * len(self.params) * 56:
            self._new_capture_arenas(self._capture_slots)
        else:
            self._capture_tables = []

    def _capture_blob(self) -> bytearray:
        blob = bytearray()
        for k, (p, gi) in enumerate(zip(self.params, self._lr_index)):
            st = self.state[p]
            blob += struct.pack('<QQQQqqfi', p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(),
                                st['exp_avg_sq'].data_ptr(), p.numel(), self._chunk0[k], float(self.param_groups[gi]['lr']),
                                self._sched_slot(gi))
        return blob

    def _table_of_this_capture(self) -> 'torch.Tensor':
        """A step captured into a graph whose backward handed over FRESH gradient tensors (``p.grad = None`` before the
        backward, what ``zero_grad(set_to_none=True)`` does: autograd then stores each gradient as it is -- no zero fill and
        no "+=" launch per parameter): the tensors live in the graph's private pool, at addresses that are the same in every
        replay of THIS graph and differ from graph to graph.  The capture therefore gets a descriptor table of its own --
        same layout and chunk space as the bound one, these gradient addresses -- uploaded by a copy node from a pinned
        host blob that the optimizer keeps alive (2 KB per replay).

        Slots: a table is keyed by its CONTENT minus the learning rates (the addresses): the pieces of a step taken in pieces
        (``step(groups=...)`` several times in one capture) and a re-capture that lands on the same addresses share one slot.
        Slots are recycled when the addresses they describe die (``release_captured_tables``; automatic on optimizer
        surgery that re-creates tensors), and the arenas are re-made when the number of parameters changes."""
        for p in self.params:
            if p.grad is None:  # no gradient reached it in this backward: a zero one from the graph's pool
                p.grad = torch.zeros_like(p)
            assert p.grad.is_contiguous() and p.grad.dtype == torch.float32 and cap_store(p) is None, \
                'captured steps with fresh gradients: contiguous float32 gradients, no row capacity'
        blob = self._capture_blob()
        n = len(blob)
        assert self._pin_arena.numel() == self._capture_slots * n, \
            'FusedAdam: the parameter list changed since the capture arenas were made; call release_captured_tables() ' \
            'outside graph capture first'
        key = bytes(b for k in range(len(self.params)) for b in blob[56 * k:56 * k + 48])  # everything but (lr, pad)
        for ent in self._capture_tables:
            if ent['key'] == key:
                return ent['table']
        used = len(self._capture_tables)
        if used >= self._capture_slots:
            raise RuntimeError(
                f'FusedAdam: {used} captured steps with gradients of their own are alive and the arenas hold '
                f'{self._capture_slots}; discard the graphs you no longer replay and call release_captured_tables() (outside '
                f'capture), or reserve more with release_captured_tables(slots=N)')
        pin, table = self._pin_arena[used * n:(used + 1) * n], self._table_arena[used * n:(used + 1) * n]
        pin.copy_(torch.frombuffer(blob, dtype=torch.uint8))
        table.copy_(pin, non_blocking=True)
        self._capture_tables.append(dict(key=key, pin=pin, table=table))
        return table

    def _refresh_captured_rates(self):
        """push the current learning rates into the tables of captured steps, ORDERED on the current stream: the copy nodes
        of a replay read the pinned blob when they execute, so a plain host store could reach replays that are queued but
        have not run yet (their result would depend on how far the device lags).  The new descriptors therefore travel
        host -> device scratch (``_h2d``: staged, event-guarded) -> pinned blob (a device-to-host copy ON the stream):
        replays enqueued before this call see the old rates, replays after it the new ones."""
        if not self._capture_tables:
            return
        n = len(self.params) * 56
        scratch = torch.empty(n, dtype=torch.uint8, device=self._table.device)
        for ent in self._capture_tables:
            blob = bytearray(ent['pin'].numpy().tobytes())  # addresses as captured; only the rates are rewritten
            for k, gi in enumerate(self._lr_index):
                struct.pack_into('<fi', blob, 56 * k + 48, float(self.param_groups[gi]['lr']), self._sched_slot(gi))
            self._h2d(scratch, blob)
            ent['pin'].copy_(scratch, non_blocking=True)
            ent['table'].copy_(scratch, non_blocking=True)

    def _h2d(self, dst: 'torch.Tensor', blob) -> None:
        """small host table -> device through a pinned staging tensor, asynchronously on the current stream (a pageable
        source makes every such copy a host synchronisation: three of them per densification)"""
        n = len(blob)
        pin = getattr(self, '_pin', None)
        if pin is None or pin.numel() < n:
            self._pin = pin = torch.empty(max(n, 4096), dtype=torch.uint8, pin_memory=True)
            self._pin_event = None
        if self._pin_event is not None:
            self._pin_event.synchronize()  # the previous copy out of the staging tensor has been issued AND has run
        pin[:n].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        dst[:n].copy_(pin[:n], non_blocking=True)
        self._pin_event = torch.cuda.Event()
        self._pin_event.record()

    def rebind(self):
        """re-read every parameter's / gradient's address (after ``ViewParallel`` moved the ``.grad`` tensors into a flat
        buffer, or before capturing ``step`` into a graph: the refresh in ``step`` cannot run during capture)"""
        self._upload()

    # ---------------------------------------------------------------------------------------------------------
    def add_state_listener(self, fn):
        """``fn()`` is called after every change of the Adam moments that does not come from a step: ``load_state_dict``,
        ``change_optimizer``, ``gather_rows``.  What depends on the moments registers here -- the live-tile mask of the sparse
        logit-table update (``FusedSuperpointStep.refresh_logit_mask``): a restored checkpoint whose live tiles the current
        neighbours do not touch would otherwise stop being updated, silently diverging from dense Adam.  A listener that returns
        False is dropped (its owner is gone)."""
        self._state_listeners.append(fn)

    def _notify_state_changed(self):
        self._state_listeners = [fn for fn in self._state_listeners if fn() is not False]

    def change_optimizer(self, tensor, name=None, op: str = 'replace', dim: int = 0) -> dict:
        """Replace, prune or extend the parameter of the named group(s) together with its Adam state -- the optimizer
        surgery of densification, ``GaussianSplatting.change_optimizer`` (networks/gaussian_splatting.py:515-563):

          replace : the group's parameter becomes ``tensor``; exp_avg / exp_avg_sq restart from zero
          prune   : ``tensor`` is a boolean keep-mask along ``dim``; parameter and state keep the selected rows
          concat  : ``tensor`` is appended along ``dim``; the new rows' state starts from zero

        ``tensor`` may be a dict name -> tensor.  Like the reference, a named group holds ONE parameter.  Returns
        name -> new ``nn.Parameter`` (the caller re-binds its module attributes, rebuilds any flat gradient buffer and
        re-captures graphs: every shape changed)."""
        from torch import nn
        assert op in ('replace', 'prune', 'concat')
        names = ([name] if isinstance(name, str) else list(name)) if name is not None else list(tensor.keys())
        out = {}
        for g in self.param_groups:
            if g.get('name') not in names:
                continue
            assert len(g['params']) == 1, f"group {g['name']!r}: change_optimizer handles one parameter per group"
            old = g['params'][0]
            new_t = tensor[g['name']] if isinstance(tensor, dict) else tensor
            if cap_store(old) is not None:  # row capacity: the same Parameter object, the same storage
                out[g['name']] = self._change_in_place(old, new_t, op, dim)
                continue
            st = self.state.pop(old)
            with torch.no_grad():
                if op == 'concat':
                    new_t = new_t.to(old.device, torch.float32)
                    data = torch.cat([old.data, new_t], dim)
                    st = {k: torch.cat([v, torch.zeros_like(new_t)], dim) for k, v in st.items()}
                elif op == 'prune':
                    idx = (slice(None),) * dim + (new_t,)
                    data = old.data[idx]
                    st = {k: v[idx] for k, v in st.items()}
                else:
                    data = new_t.detach().to(old.device, torch.float32)
                    st = {k: torch.zeros_like(data) for k in st}
            p = nn.Parameter(data.contiguous().requires_grad_(True))
            g['params'][0] = p
            self.state[p] = {k: v.contiguous() for k, v in st.items()}
            out[g['name']] = p
        self.params, self._lr_index = [], []
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if p.requires_grad:
                    self.params.append(p)
                    self._lr_index.append(gi)
        if self._table.numel() != len(self.params) * 56:
            self._table = torch.zeros(len(self.params) * 56, dtype=torch.uint8, device=self._table.device)
        self._upload()
        self._addresses_changed()
        self._notify_state_changed()
        return out

    @torch.no_grad()
    def gather_rows(self, names, rows: 'torch.Tensor', n_keep: int) -> dict:
        """Rebuild the parameters of the named groups and their Adam moments as row gathers, ALL in one launch
        (csrc/densify.hip::skgs_gather_rows): row i of every new tensor is row ``rows[i]`` of the old one; rows
        ``i >= n_keep`` are new Gaussians -- parameters copied from their parent, moments zero (gaussian_splatting.py:
        531-545).  Prune = the kept rows, clone = all rows + the selected ones, split = the unselected rows + N copies of
        the selected ones (:565-634).  Returns name -> new ``nn.Parameter`` like ``change_optimizer``."""
        from torch import nn
        lib = _C.load_library()
        lib.skgs_row_tensor_bytes.restype = C.c_size_t
        assert int(lib.skgs_row_tensor_bytes()) == 24
        rows = rows.to(torch.int64).contiguous()
        n_out, dev = int(rows.numel()), rows.device
        named = [g for g in self.param_groups if g.get('name') in names]
        if named and all(cap_store(g['params'][0]) is not None for g in named):
            if all(n_out <= cap_store(g['params'][0]).shape[0] for g in named):
                out = self._gather_rows_in_place(named, rows, n_out, int(n_keep))
                self._notify_state_changed()
                return out
            raise CapacityExceeded(f'{n_out} rows do not fit the row capacity {cap_store(named[0]["params"][0]).shape[0]}')
        blob, new, max_rf, keepalive = bytearray(), {}, 1, []
        for g in self.param_groups:
            if g.get('name') not in names:
                continue
            assert len(g['params']) == 1, f"group {g['name']!r}: one parameter per group"
            old = g['params'][0]
            st = self.state.pop(old)
            rf = old.numel() // max(old.shape[0], 1)
            max_rf = max(max_rf, rf)
            shape = (n_out,) + tuple(old.shape[1:])
            p = nn.Parameter(torch.empty(shape, dtype=torch.float32, device=dev), requires_grad=True)
            m, v = torch.empty(shape, dtype=torch.float32, device=dev), torch.empty(shape, dtype=torch.float32, device=dev)
            for src, dst, fresh_zero in ((old.data, p.data, 0), (st['exp_avg'], m, 1), (st['exp_avg_sq'], v, 1)):
                assert src.is_contiguous()
                blob += struct.pack('<QQii', src.data_ptr(), dst.data_ptr(), rf, fresh_zero)
            keepalive += [old, st]
            g['params'][0] = p
            self.state[p] = dict(exp_avg=m, exp_avg_sq=v)
            new[g['name']] = p
        if new:
            table = torch.empty(len(blob), dtype=torch.uint8, device=dev)
            self._h2d(table, blob)
            _C._check(lib.skgs_gather_rows(C.c_int32(3 * len(new)), C.c_void_p(table.data_ptr()), C.c_int64(n_out),
                                           C.c_int64(int(n_keep)), C.c_void_p(rows.data_ptr()), C.c_int32(max_rf),
                                           _C._stream()))
        self.params, self._lr_index = [], []
        for gi, g in enumerate(self.param_groups):
            for q in g['params']:
                if q.requires_grad:
                    self.params.append(q)
                    self._lr_index.append(gi)
        if self._table.numel() != len(self.params) * 56:
            self._table = torch.zeros(len(self.params) * 56, dtype=torch.uint8, device=self._table.device)
        self._upload()
        self._addresses_changed()
        del keepalive  # (the old tensors lived until the launch was enqueued: same stream, the allocator orders reuse)
        self._notify_state_changed()
        return new

    @torch.no_grad()
    def _change_in_place(self, p, new_t, op: str, dim: int):
        """``change_optimizer`` for a parameter with a row capacity (rows are dimension 0)"""
        assert dim == 0, 'a row capacity grows along dimension 0'
        store, (m_store, v_store) = cap_store(p), self._cap_state[p]
        n0 = p.shape[0]
        if op == 'replace':
            new_t = new_t.detach().to(p.device, torch.float32)
            n1 = new_t.shape[0]
            if n1 > store.shape[0]:
                raise CapacityExceeded(f'{n1} rows do not fit the row capacity {store.shape[0]}')
            store[:n1].copy_(new_t)
            m_store[:n1].zero_(), v_store[:n1].zero_()
        elif op == 'concat':
            new_t = new_t.detach().to(p.device, torch.float32)
            n1 = n0 + new_t.shape[0]
            if n1 > store.shape[0]:
                raise CapacityExceeded(f'{n1} rows do not fit the row capacity {store.shape[0]}')
            store[n0:n1].copy_(new_t)
            m_store[n0:n1].zero_(), v_store[n0:n1].zero_()
        else:  # prune: a keep-mask over the rows
            rows = torch.nonzero(new_t).squeeze(1)
            n1 = int(rows.numel())
            for home in (store, m_store, v_store):
                home[:n1].copy_(home.index_select(0, rows))
        p.data = store[:n1]
        self.state[p] = dict(exp_avg=m_store[:n1], exp_avg_sq=v_store[:n1])
        regrad(p)
        return p

    @torch.no_grad()
    def _gather_rows_in_place(self, named, rows, n_out: int, n_keep: int) -> dict:
        """``gather_rows`` for parameters with a row capacity: the same ONE gather launch, into scratch tensors, then the rows
        go back into the SAME storage (parameters and both moments) and every view -- ``p.data``, the moments, ``p.grad`` --
        is re-shaped over it.  No address changes: a captured step keeps replaying (sk_gs_amd/capacity.py)."""
        lib = _C.load_library()
        dev = rows.device
        blob, max_rf, scratch, homes = bytearray(), 1, [], []
        for g in named:
            p = g['params'][0]
            st, (m_store, v_store) = self.state[p], self._cap_state[p]
            rf = p.numel() // max(p.shape[0], 1)
            max_rf = max(max_rf, rf)
            shape = (n_out,) + tuple(p.shape[1:])
            for src, home, fresh_zero in ((p.data, cap_store(p), 0), (st['exp_avg'], m_store, 1), (st['exp_avg_sq'], v_store, 1)):
                tmp = torch.empty(shape, dtype=torch.float32, device=dev)
                blob += struct.pack('<QQii', src.data_ptr(), tmp.data_ptr(), rf, fresh_zero)
                scratch.append(tmp)
                homes.append(home[:n_out])
        table = torch.empty(len(blob), dtype=torch.uint8, device=dev)
        self._h2d(table, blob)
        _C._check(lib.skgs_gather_rows(C.c_int32(len(scratch)), C.c_void_p(table.data_ptr()), C.c_int64(n_out),
                                       C.c_int64(n_keep), C.c_void_p(rows.data_ptr()), C.c_int32(max_rf), _C._stream()))
        torch._foreach_copy_(homes, scratch)
        out = {}
        for g in named:
            p = g['params'][0]
            m_store, v_store = self._cap_state[p]
            p.data = cap_store(p)[:n_out]
            self.state[p] = dict(exp_avg=m_store[:n_out], exp_avg_sq=v_store[:n_out])
            regrad(p)
            out[g['name']] = p
        self._upload()  # the live element counts of the descriptor table (same pointers, same chunk ranges)
        return out

    def set_lr(self, group, lr: float):
        """``group``: index or name of the parameter group.  Re-uploads the descriptor table (outside graph capture); a
        captured ``step`` reads the table at every replay, so the new rate applies to replays too."""
        if isinstance(group, str):
            group = next(i for i, g in enumerate(self.param_groups) if g.get('name') == group)
        self.param_groups[group]['lr'] = float(lr)
        self._upload()
        self._refresh_captured_rates()  # tables of captured steps (_table_of_this_capture), stream-ordered

    # ------------------------------------------------------------------------------------------ device learning-rate schedules
    def _sched_slot(self, gi: int) -> int:
        """the `sched` word of a descriptor: 0, or 1 + the schedule of the parameter's group"""
        return self._sched_of_group.get(gi, -1) + 1

    def set_lr_schedule(self, groups, lr_init: float, lr_final: float, max_steps: int, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
                        step_offset: int = 0):
        """The named group(s) follow ``get_expon_lr_func(lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps)`` (networks/
        gaussian_splatting.py:56-84) of the 1-based training step minus ``step_offset`` -- evaluated ON THE DEVICE by the launch that
        advances the step counter (``skgs_adam_set_lr_schedules``, csrc/adam_update.h): what the reference's ``update_learning_rate``
        does from a before-train-step hook (train.py:140-141; `xyz`: gaussian_splatting.py:455-470; the deform networks' groups with
        the stage's first step as offset: sk_gs.py:611-632), without the host -- a step replayed in a hipGraph, several steps per
        replay, follows it step for step.  Up to 4 schedules; outside graph capture.  ``clear_lr_schedules()`` returns to ``set_lr``."""
        names = [groups] if isinstance(groups, (str, int)) else list(groups)
        idx = [n if isinstance(n, int) else next(i for i, g in enumerate(self.param_groups) if g.get('name') == n) for n in names]
        entry = (float(lr_init), float(lr_final), float(lr_delay_mult), int(lr_delay_steps), int(max_steps), int(step_offset))
        # the table is REBUILT from the entries some group still follows (appending only, a re-scheduled group's old entry
        # kept its slot; the reference's piecewise schedule -- offset 0, then sp_fix[0], then sk_init[0] for `xyz` and for the deform
        # groups, sk_gs.py:619-626: six distinct entries over a run -- hit the limit at the second stage boundary).  A device schedule
        # carries ONE fixed step_offset: the host re-sets it at every stage boundary, as update_learning_rate recomputes its offset.
        by_group = {gi: self._schedules[slot] for gi, slot in self._sched_of_group.items()}
        for gi in idx:
            by_group[gi] = entry
        table = []
        for e in by_group.values():
            if e not in table:
                table.append(e)
        assert len(table) <= 4, 'FusedAdam: at most 4 device learning-rate schedules in use at a time'
        self._schedules = table
        self._sched_of_group = {gi: table.index(e) for gi, e in by_group.items()}
        self._upload_schedules()

    def clear_lr_schedules(self):
        self._schedules, self._sched_of_group = [], {}
        self._upload_schedules()

    def _upload_schedules(self):
        lib = _C.load_library()
        import numpy as np
        lg = lambda v: float(np.log(v)) if v > 0 else 0.0  # noqa: E731  (np.log: what get_expon_lr_func itself evaluates)
        blob = b''.join(struct.pack('<dddddiiii', a, b, c, lg(a), lg(b), d, e, f, 0) for a, b, c, d, e, f in self._schedules)
        n = len(self._schedules)
        if n:  # (a new tensor per upload: launches already queued keep reading the old array)
            self._sched_dev = torch.empty(max(len(blob), 64), dtype=torch.uint8, device=self._table.device)
            self._h2d(self._sched_dev, blob)
        _C._check(lib.skgs_adam_set_lr_schedules(C.c_void_p(self.step_state.data_ptr()), C.c_void_p(self._sched_dev.data_ptr() if n else None),
                                                 C.c_int32(n), _C._stream()))
        self._upload()
        self._refresh_captured_rates()

    def scheduled_lr(self, group) -> float:
        """(synchronising) the rate the next step applies to ``group``: its schedule's current value on the device, or its ``lr``"""
        gi = group if isinstance(group, int) else next(i for i, g in enumerate(self.param_groups) if g.get('name') == group)
        slot = self._sched_of_group.get(gi)
        if slot is None:
            return float(self.param_groups[gi]['lr'])
        return float(self.step_state[8 + slot].item())   # AdamState.lr_now (byte 32)

    # ---------------------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        """``torch.optim.Adam.state_dict()`` layout: ``state`` = {param index: {step, exp_avg, exp_avg_sq}} and
        ``param_groups`` with ``params`` as index lists (+ ``lr``, ``name``, ``betas``, ``eps``): what the reference's
        framework saves next to the model (my_ext/framework.py checkpointing) -- moments AND the step counter, so that a
        resumed run continues the bias correction instead of restarting it.  (Synchronises: reads the device counter.)"""
        index = {p: i for i, p in enumerate(self.params)}
        step = float(self.step_count.item())
        state = {i: dict(step=torch.tensor(step), exp_avg=self.state[p]['exp_avg'].detach().clone(),
                         exp_avg_sq=self.state[p]['exp_avg_sq'].detach().clone()) for p, i in index.items()}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d.update(betas=tuple(self.betas), eps=self.eps, amsgrad=False, weight_decay=0)
            d['params'] = [index[p] for p in g['params'] if p in index]
            groups.append(d)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd: dict):
        """inverse of ``state_dict`` (also accepts one saved by ``torch.optim.Adam`` over the same groups): moments by
        parameter index, learning rates by group, the step counter from the entries' ``step`` (they must agree)"""
        groups = sd['param_groups']
        assert len(groups) == len(self.param_groups), 'load_state_dict: different number of parameter groups'
        for g, saved in zip(self.param_groups, groups):
            assert len(saved['params']) == len([p for p in g['params'] if p.requires_grad]), 'group sizes differ'
            if 'lr' in saved:
                g['lr'] = float(saved['lr'])
        steps = set()
        with torch.no_grad():
            for i, p in enumerate(self.params):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is None:  # torch omits parameters that never received a gradient
                    self.state[p]['exp_avg'].zero_(), self.state[p]['exp_avg_sq'].zero_()
                    continue
                assert tuple(st['exp_avg'].shape) == tuple(p.shape), f'parameter {i}: shape differs'
                self.state[p]['exp_avg'].copy_(st['exp_avg'])
                self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
                steps.add(float(st['step']))
            assert len(steps) <= 1, f'load_state_dict: the parameters disagree on the step count {sorted(steps)} ' \
                                    '(the fused kernel keeps ONE counter)'
            self._set_step_count(steps.pop() if steps else 0.0)
        self._upload()
        self._refresh_captured_rates()  # captured steps keep their addresses; the restored rates reach them too
        self._notify_state_changed()

    def _set_step_count(self, count: float):
        """restore a step count: the counter and the two bias-correction terms 1 - beta^count the kernels advance by
        recurrence (doubles, as torch forms them)"""
        blob = struct.pack('<ffdd', float(count), 0.0, 1.0 - self.betas[0] ** count, 1.0 - self.betas[1] ** count)
        self.step_state.zero_()
        self._h2d(self.step_state.view(torch.uint8), blob)
        if self._schedules:  # (the state also holds the schedules' pointer and current rates: re-derived for the restored count)
            lib = _C.load_library()
            _C._check(lib.skgs_adam_set_lr_schedules(C.c_void_p(self.step_state.data_ptr()), C.c_void_p(self._sched_dev.data_ptr()),
                                                     C.c_int32(len(self._schedules)), _C._stream()))

    def zero_grad(self, set_to_none: bool = False):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def _chunk_ranges(self, groups):
        """chunk ranges of the named groups' parameters: one per run of neighbours in the table"""
        groups = set(groups)
        known = {g.get('name') for g in self.param_groups}
        assert groups <= known, f'unknown parameter groups {sorted(groups - known)}'
        idx = [i for i, gi in enumerate(self._lr_index) if self.param_groups[gi].get('name') in groups]
        runs = []
        for i in idx:
            if runs and runs[-1][1] == self._chunk0[i]:
                runs[-1][1] = self._chunk0[i + 1]
            else:
                runs.append([self._chunk0[i], self._chunk0[i + 1]])
        return [tuple(r) for r in runs if r[1] > r[0]]

    def step(self, groups=None, advance: bool = True):
        """One Adam step over every parameter, or -- ``groups`` = names of parameter groups -- the piece of it that
        updates those groups (one launch per run of groups that are neighbours in the table).  A step taken in pieces lets each piece start as soon as ITS gradients are final, on a
        side stream beside the rest of the backward (``OverlappedStep``): every piece uses the bias correction of the
        same step; pass ``advance=False`` to all of them and call ``advance_step()`` once, after they have all been
        ordered before it (the counter every piece reads must not move under them).

        After ``FusedTrainStep.loss(...).backward()`` the call is that step's TAIL: the per-Gaussian rows were updated by the
        backward's skeleton launch, what is left is the closing launch."""
        pending = self._pending_tail
        if pending is not None and groups is None and advance:
            self._pending_tail = None
            pending[0]._tail(pending[1])
            return
        table = self._table
        # the table holds raw gradient pointers: autograd may have replaced a .grad tensor (zero_grad(set_to_none=True))
        if any(p.grad is None or p.grad.data_ptr() != g for p, g in zip(self.params, self._bound_grads)):
            if torch.cuda.is_current_stream_capturing():
                table = self._table_of_this_capture()
            else:
                self._upload()
        lib = _C.load_library()
        z = self.zero_after_step if advance else None
        ranges = [(0, self._total_chunks)] if groups is None else self._chunk_ranges(groups)
        if advance and not ranges:
            ranges = [(0, 0)]
        for k, (c0, c1) in enumerate(ranges):
            last = advance and k == len(ranges) - 1
            _C._check(lib.skgs_adam_step_range(
                C.c_int32(len(self.params)), C.c_void_p(table.data_ptr()), C.c_int64(c0), C.c_int64(c1),
                C.c_double(self.betas[0]), C.c_double(self.betas[1]), C.c_double(self.eps),
                C.c_void_p(self.step_state.data_ptr()), C.c_int32(1 if last else 0),
                C.c_void_p(z.data_ptr() if (z is not None and last) else None),
                C.c_int64(z.numel() if (z is not None and last) else 0), _C._stream()))

    def side_range(self, groups, part=None, after_advance: bool = False) -> AdamRange:
        """the piece of a step that updates ``groups`` (neighbours in the table) as a ``skgs_adam_range``: handed to
        ``skgs_deform_mlp_backward_adam`` / ``skgs_skeleton_backward`` / ``skgs_skeleton_forward`` it runs on the CUs that
        launch leaves idle -- same arithmetic as ``step(groups, advance=False)``.  ``part`` = (lo, hi) fractions: only that
        share of the chunks (two launches can split one run); ``after_advance``: the piece is issued after the launch that
        advanced the counter but belongs to the step that launch closed."""
        ranges = self._chunk_ranges(groups)
        assert len(ranges) == 1, f'groups {list(groups)} are not one run of neighbours in the table'
        c0, c1 = ranges[0]
        if part is not None:
            n = c1 - c0
            c0, c1 = c0 + int(round(n * part[0])), c0 + int(round(n * part[1]))
        return AdamRange(len(self.params), self._table.data_ptr(), c0, c1, self.betas[0], self.betas[1],
                         self.eps, self.step_state.data_ptr(), int(after_advance))

    def table_entry(self, param) -> int:
        """device address of ``param``'s descriptor in the optimizer's table (param, grad, exp_avg, exp_avg_sq, n, chunk0, lr):
        what the special-purpose update launches read their tensor from (``skgs_adam_logit_rows``), so that ``set_lr`` and
        re-homed storage reach them like every other piece of the step"""
        i = next(k for k, q in enumerate(self.params) if q is param)
        return self._table.data_ptr() + 56 * i

    def step_tail(self, groups, freq_job=None, freq_param=None, next_view=None):
        """the closing piece of a step: ``groups`` (neighbours in the table) are updated, the counter advances,
        ``zero_after_step`` is cleared -- ``skgs_adam_step_tail``.  ``freq_job`` = (B, D, degree, grad_out, out, ld_out,
        grad_x, accumulate) with device tensors: the frequency-encoding backward that completes ``freq_param``'s gradient
        (the joints': their gradient through the network input) runs first, inside the workgroup that updates that
        (one-chunk) tensor -- no launch between the backward and the update.  ``next_view`` (``ViewTable.advance()``): the
        launch ends by putting the next training view's record into the view slot."""
        lib = _C.load_library()
        ranges = self._chunk_ranges(groups)
        assert len(ranges) == 1, f'groups {list(groups)} are not one run of neighbours in the table'
        z = self.zero_after_step
        if freq_job is None:
            chunk, fj = -1, (0, 1, 0, None, None, 0, None, 0)
        else:
            B, D, deg, g, out, ld, gx, acc = freq_job
            i = next(k for k, q in enumerate(self.params) if q is freq_param)
            assert self._chunk0[i + 1] - self._chunk0[i] == 1 and gx.data_ptr() == freq_param.grad.data_ptr()
            chunk, fj = self._chunk0[i], (B, D, deg, g.data_ptr(), out.data_ptr(), ld, gx.data_ptr(), int(acc))
        _C._check(lib.skgs_adam_step_tail(
            C.c_int32(len(self.params)), C.c_void_p(self._table.data_ptr()), C.c_int64(ranges[0][0]), C.c_int64(ranges[0][1]),
            C.c_double(self.betas[0]), C.c_double(self.betas[1]), C.c_double(self.eps), C.c_void_p(self.step_state.data_ptr()),
            C.c_void_p(z.data_ptr() if z is not None else None), C.c_int64(z.numel() if z is not None else 0),
            C.c_int64(chunk), C.c_int32(fj[0]), C.c_int32(fj[1]), C.c_int32(fj[2]), C.c_void_p(fj[3]), C.c_void_p(fj[4]),
            C.c_int32(fj[5]), C.c_void_p(fj[6]), C.c_int32(fj[7]), None if next_view is None else C.byref(next_view),
            _C._stream()))

    def advance_step(self):
        """close a step taken in pieces: the counter moves, ``zero_after_step`` is cleared"""
        lib = _C.load_library()
        z = self.zero_after_step
        _C._check(lib.skgs_adam_step_range(C.c_int32(0), None, C.c_int64(0), C.c_int64(0), C.c_double(self.betas[0]),
                                           C.c_double(self.betas[1]), C.c_double(self.eps),
                                           C.c_void_p(self.step_state.data_ptr()), C.c_int32(1),
                                           C.c_void_p(z.data_ptr() if z is not None else None),
                                           C.c_int64(z.numel() if z is not None else 0), _C._stream()))
