"""The per-view training step as a straight line of C-ABI calls on persistent buffers.

``SkinnedGaussians.render`` + ``image_loss`` + ``loss.backward()`` run the hot path through torch autograd: every
operator of include/skgs.h is one autograd node, and autograd adds its own kernels around them -- ``torch.cat`` of
the two SH parameters and the split of its gradient, gather / softmax / scatter for the LBS logits, index-select of the
per-frame tables and the zero-filled tables of their backward, one ``AccumulateGrad`` add per parameter, zero fills for
every freshly allocated gradient.  On one MI355X (config #1) that glue is ~40 short launches and ~0.2 ms of a 0.85 ms
step.  ``FusedViewStep`` issues the SAME library calls in the same order, but

  * reads the parameters in place (split SH storage: ``skgs_raster_inputs.sh_rest``; per-frame rows by pointer),
  * composites the background inside the blend kernels (``skgs_raster_inputs.background``),
  * lets every backward kernel write its result straight into the parameter's ``.grad`` storage (the views of
    ``view_parallel.FlatGradBuffer``), overwriting instead of accumulating,

so a step is ~22 kernels of ours and one tiny fill, with no autograd graph at all.  It computes exactly what the autograd
path computes (tests/test_gpu_fused_step.py compares every gradient); the autograd path stays the drop-in operator
surface.

Reference call sequence: networks/sk_gs.py:1160-1242 (forward / render), :1524-1529 (loss), train.py:179-250.
"""
import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

from sk_gs_amd import _C
from sk_gs_amd.model import SkinnedGaussians
from sk_gs_amd.renderer.gaussian_render import GaussianRasterizationSettings


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(None if t is None else t.data_ptr())


class FusedViewStep:
    """forward + loss + backward of one view for a ``SkinnedGaussians`` model (stage ``sk``, sp_W weights).

    After ``forward_backward(rs, time_id, target)``:
      * every parameter's ``.grad`` holds the gradient of ``lambda_l1 * L1 + lambda_ssim * (1 - SSIM)`` (overwritten),
      * ``loss3`` (device, 3 floats) = total, L1 mean, SSIM mean; ``grad_means2D`` [P,3] is the screen-space gradient
        the densification statistics use (``viewspace_points.grad``, gaussian_splatting.py:503-513);
      * ``image`` [3,H,W], ``out_opacity`` [H,W], ``radii`` [P] are the forward results.
    No host synchronisation: the binning buffer has a fixed ``capacity`` (tile instances); ``status()`` reads the
    device-side overflow flag.  hipGraph-capturable.
    """

    def __init__(self, model: SkinnedGaussians, W: int, H: int, capacity: int, lambda_dssim: float = 0.2,
                 background: Optional[Tensor] = None, grad_scale: float = 1.0, densify_stats: bool = False,
                 spw_logit_grad: Optional[Tensor] = None, tables_zeroed_by_optimizer: bool = False,
                 tile_bucket: int = 0, sh_factors: Optional[Tensor] = None, fused_deform_net: bool = True,
                 view_table=None):
        assert not model.static, 'FusedViewStep covers the skinned stage (M >= 1)'
        self.model, self.W, self.H = model, int(W), int(H)
        self.lambda_l1, self.lambda_ssim = 1.0 - lambda_dssim, lambda_dssim
        lib = self.lib = _C.load_library()
        dev = model._xyz.device
        if not model._xyz.is_cuda:
            raise _C.SkgsError('FusedViewStep needs the model on a HIP device; sk_gs_amd has no CPU path')
        # With a row capacity (model.enable_capacity, sk_gs_amd/capacity.py) every per-Gaussian buffer and launch is sized for
        # the CAPACITY and the kernels read the live count from a device word: the step -- and a hipGraph of it -- survives
        # clone / split / prune.  `self.P` is what the launches are sized for; `model.P` is the live count.
        cap = getattr(model, 'capacity', None)
        P, M, K = (cap.P_cap if cap is not None else model.P), model.M, model.K
        self.P, self.M, self.K = P, M, K
        self._live = cap.live if cap is not None else None
        assert cap is None or (sh_factors is None and spw_logit_grad is None), \
            'row capacity: one-rank step (the factor / compact-logit exchanges are sized by the live count)'
        # one limit for every one-launch skinning path (the library's): beyond it -- the 512 superpoints of the sp stage --
        # the step uses the separate KNN / weights / skinning launches, which have no bone limit
        lib.skgs_fused_lbs_max_bones.restype = C.c_int
        self.max_fused_bones = int(lib.skgs_fused_lbs_max_bones())
        self.wide = M > self.max_fused_bones or K > 8
        # the one-launch skinning as a job of the rasterizer's per-Gaussian launch (skgs_raster_inputs.deform_job): the same
        # bits one launch earlier; SKGS_SEPARATE_DEFORM=1 keeps the two launches (A/B measurements)
        self.deform_in_preprocess = os.environ.get('SKGS_SEPARATE_DEFORM', '0') != '1'
        # likewise the skinning backward (moments path: M <= 64, K <= 8) as a job of the rasterizer's per-Gaussian backward
        # launch (skgs_raster_grads.deform_backward_job); SKGS_SEPARATE_DEFORM_BACKWARD=1 keeps the launch of its own
        self.deform_backward_in_preprocess = os.environ.get('SKGS_SEPARATE_DEFORM_BACKWARD', '0') != '1'
        self._rows_backward_done = False
        # the weighting of calc_LBS_weight: `W` (logits per Gaussian) runs inside the one-launch skinning kernels; the three
        # distance-based ones (sk_gs.py:757-766,770) as search + weighting in one launch, then the skinning
        self.lbs_method = getattr(model, 'lbs_method', 'W')
        assert self.lbs_method == 'W' or (cap is None and spw_logit_grad is None), \
            'the distance-based LBS weightings run without a row capacity / compact-logit exchange'
        if K > 16 or K > M:
            raise _C.SkgsError(f'FusedViewStep: K = {K} neighbours: the KNN kernels keep at most 16 (and K <= M = {M})')
        # view-parallel training: instead of the [P,16,3] SH gradient of this view, write its two factors here ([P,6]:
        # unit view direction, clamp-masked colour gradient); ``sh_grads_from_factors`` rebuilds the rows of ALL views
        # from the all-gathered factors (24 bytes per Gaussian and view on the wire instead of 192 all-reduced)
        self.sh_factors = sh_factors
        assert sh_factors is None or (sh_factors.is_cuda and sh_factors.dtype == torch.float32 and
                                      sh_factors.is_contiguous() and sh_factors.numel() == P * 6)
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        self.background = None if background is None else background.to(**f32).contiguous()
        # dL/dloss seed: 1/world_size makes the gradients arrive pre-averaged for a SUM all-reduce (view-parallel
        # training), instead of a separate division pass over the whole flat gradient buffer
        self.grad_scale = None if grad_scale == 1.0 else torch.full((1,), float(grad_scale), **f32)
        self._grad_scale_value = float(grad_scale)
        from sk_gs_amd.capacity import cap_store
        for p in model.parameters():
            if p.grad is None:
                store = cap_store(p)
                if store is None:
                    p.grad = torch.zeros_like(p)
                else:  # room for the whole capacity behind the live rows
                    p.grad, p._grad_slot = torch.zeros_like(store)[:p.shape[0]], store.numel()
            assert cap_store(p) is None or getattr(p, '_grad_slot', 0) >= cap_store(p).numel(), \
                'row capacity: build the gradient buffers AFTER model.enable_capacity()'
            assert p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
        # ---- persistent intermediates -------------------------------------------------------------------------
        self.bone_T, self.chain_A = torch.empty((M, 7), **f32), torch.empty((M, 7), **f32)
        self.indices = torch.empty((P, K), dtype=torch.int64, device=dev)
        self.weights = torch.empty((P, K), **f32)
        self.means, self.scales = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        self.rotations, self.opacity = torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
        self.image, self.out_opacity = torch.empty((3, H, W), **f32), torch.empty((H, W), **f32)
        self.radii = torch.empty((P,), dtype=torch.int32, device=dev)
        self.geom = torch.empty((lib.skgs_geom_buffer_bytes(C.c_int32(P)),), **u8)
        self.geom[:256].zero_()  # status words; `overflow_events` counts overflowing forwards from here on
        self.img = torch.empty((lib.skgs_img_buffer_bytes(C.c_int32(W), C.c_int32(H)),), **u8)
        # tile_bucket = Lcap > 0: every tile owns Lcap fixed slots (skgs_raster_inputs.tile_bucket_capacity): no counting
        # and no scan launch; `capacity` is then ignored
        self.tile_bucket = int(tile_bucket)
        if self.tile_bucket > 0:
            capacity = ((W + 15) // 16) * ((H + 15) // 16) * self.tile_bucket
        self.binning = torch.empty((lib.skgs_binning_buffer_bytes(C.c_int64(int(capacity))),), **u8)
        self.loss3 = torch.zeros(3, **f32)
        self.loss_ws = torch.empty((lib.skgs_image_loss_workspace_bytes(C.c_int32(3), C.c_int32(H), C.c_int32(W)),), **u8)
        self.dL_dimage = torch.empty((3, H, W), **f32)
        self.grad_means2D = torch.empty((P, 3), **f32)
        self.g_colors, self.g_cov3D = torch.empty((P, 3), **f32), torch.empty((P, 6), **f32)
        self.g_means, self.g_scales = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        self.g_rotations, self.g_opacity = torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
        self.g_weights = torch.empty((P, K), **f32)
        self.g_bone_T = torch.empty((M, 7), **f32)
        if self.lbs_method != 'W':
            self.nn_dist = torch.empty((P, K), **f32)
            lib.skgs_knn_dist_weights_workspace_bytes.restype = C.c_size_t
            self.dist_ws = torch.empty((max(int(lib.skgs_knn_dist_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(3))),
                                            16),), **u8)
        self.deform_ws = torch.empty((lib.skgs_lbs_deform_backward_workspace_bytes(C.c_int32(P), C.c_int32(M)),), **u8)
        self.bwd_ws = torch.zeros((lib.skgs_backward_workspace_bytes(C.c_int32(P)),), **u8)  # kept zero between steps
        # the optimizer clears the per-frame table gradients after its update (FusedAdam(zero_after_step=
        # step.table_grad_span())): no fill launch at the start of the step
        self.tables_zeroed_by_optimizer = bool(tables_zeroed_by_optimizer)
        # (FusedAdam, group names) or None: that piece of the optimizer step runs inside the deform network's backward
        # launch (train_step.FusedTrainStep sets it; one rank, no gradient exchange between backward and update)
        self.side_optimizer = None
        self.keep_raster_grads = bool(os.environ.get('SKGS_KEEP_RASTER_GRADS'))  # (True: g_means / g_scales / g_rotations / g_opacity / g_colors / g_cov3D are written even when a job consumes them)
        # the skeleton stage of the view in the slot already ran (``skeleton_forward``: FusedTrainStep(pre_forward=True)
        # issues it for the NEXT view behind the optimizer's closing launch): ``forward`` starts at the skinning
        self.skeleton_ahead = False
        self.defer_input_grad = False
        self._arm_tail = None   # (FusedTrainStep.loss: what the node's backward arms on the optimizer)
        # view-parallel training: [P*K] float32 view that receives the compact LBS-logit gradient (see backward_skinning)
        self.spw_logit_grad = spw_logit_grad
        assert spw_logit_grad is None or (spw_logit_grad.numel() == P * K and spw_logit_grad.is_contiguous())
        # densification statistics (gaussian_splatting.py:503-513, sk_gs.py:1990-1997), updated by every step if asked
        self.densify_stats = bool(densify_stats)
        self._acc_store, self._den_store = torch.zeros((P, 1), **f32), torch.zeros((P, 1), **f32)
        self._rad_store = torch.zeros((P,), **f32)
        # bone-transform producer (scope row (f)-3): the MLP runs inside the step, its weight gradients are written in place
        self.deform_net = model.sk_deform_net
        self._mlp_fused = None
        if self.deform_net is not None:
            from sk_gs_amd.deform_net import DeformMLPRunner, FusedDeformMLP, fused_supported
            net = self.deform_net.dynamic_net
            # one persistent launch per direction (csrc/mlp_fused.hip) where the shape allows, else one launch per layer
            self._mlp_fused = FusedDeformMLP(self.deform_net, M) if (fused_deform_net and fused_supported(self.deform_net, M)) else None
            self._mlp = DeformMLPRunner(self.deform_net)
            if self._mlp_fused is None:
                self._x0 = torch.empty((M, net.in_channels), **f32)
                self._acts = torch.empty((net.num_layers, M, net.dim_hidden), **f32)
                self._g_act = torch.empty((2, M, net.dim_hidden), **f32)
            self._sk_r_raw, self._d_rot, self._d_scale = (torch.empty((M, 4), **f32), torch.empty((M, 4), **f32),
                                                          torch.empty((M, 3), **f32))
            self._g_heads = [torch.empty((M, 4), **f32), torch.empty((M, 4), **f32), torch.empty((M, 3), **f32)]
            self._g_x0 = torch.empty((M, net.in_channels), **f32) if getattr(model, 'learn_joints', False) else None
        # per-view inputs as device loads (sk_gs_amd/view_slot.py): forward_backward() without arguments then trains the
        # view selected in the table, and ONE captured graph serves all of them
        self.view_table = view_table
        if view_table is not None:
            assert self.deform_net is not None, 'the view slot drives the deform network\'s time input: tables mode keeps one graph per view'
            vs = view_table.settings
            assert (vs.image_height, vs.image_width) == (self.H, self.W)
        topo = model.topology()
        self._topo = topo
        self._bufs = _C._buffers(self.geom, self.binning, self.img)

    # ---- densification statistics: [P,1], [P,1], [P] over the live rows (storage of capacity rows behind them) ---------
    def _live_rows(self) -> int:
        return int(self.model.P) if self._live is not None else int(self._acc_store.shape[0])

    def _stat_get(self, store):
        return store[:self._live_rows()]

    def _stat_set(self, name, value):
        store = getattr(self, name)
        if self._live is None:  # no capacity: re-bind, as the reference re-creates its tensors
            setattr(self, name, value)
            return
        store[:value.shape[0]].copy_(value)  # capacity: the kernels (and a captured graph) keep the address

    xyz_gradient_accum = property(lambda self: self._stat_get(self._acc_store),
                                  lambda self, v: self._stat_set('_acc_store', v))
    denom = property(lambda self: self._stat_get(self._den_store), lambda self, v: self._stat_set('_den_store', v))
    max_radii2D = property(lambda self: self._stat_get(self._rad_store), lambda self, v: self._stat_set('_rad_store', v))

    @torch.no_grad()
    def reset_densify_stats(self):
        """zero the three statistics (after a densification, gaussian_splatting.py:584-587) without moving them"""
        if self._live is None:  # no capacity: new tensors of the model's new size, as the reference re-creates them
            n, dev = int(self.model.P), self._acc_store.device
            self._acc_store, self._den_store = torch.zeros((n, 1), device=dev), torch.zeros((n, 1), device=dev)
            self._rad_store = torch.zeros((n,), device=dev)
            return
        self._acc_store.zero_(), self._den_store.zero_(), self._rad_store.zero_()

    @torch.no_grad()
    def gather_densify_stats(self, rows: Tensor):
        """keep the statistics of the rows a prune keeps (gaussian_splatting.py:573-575), in place"""
        n = int(rows.numel())
        if self._live is None:
            self._acc_store, self._den_store = self._acc_store.index_select(0, rows), self._den_store.index_select(0, rows)
            self._rad_store = self._rad_store.index_select(0, rows)
            return
        for store in (self._acc_store, self._den_store, self._rad_store):
            kept = store.index_select(0, rows)
            store.zero_()
            store[:n].copy_(kept)

    # ------------------------------------------------------------------------------------------------------------
    def table_grad_span(self) -> Optional[Tensor]:
        """the per-frame tables' gradients as ONE contiguous float32 view (they are adjacent in a flat gradient buffer),
        or None when they are not adjacent"""
        m = self.model
        tabs = [t.grad for t in (m.sk_r, m.sk_d_rot, m.sk_d_scale, m.global_tr) if t is not None]
        lo = min(t.data_ptr() for t in tabs)
        total = sum(t.numel() for t in tabs)
        first = min(tabs, key=lambda t: t.data_ptr())
        same = all(t.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() for t in tabs)
        if same and max(t.data_ptr() + t.numel() * 4 for t in tabs) - lo == total * 4:
            return torch.as_strided(first, (total,), (1,))
        return None

    @torch.no_grad()
    def _zero_table_grads(self):
        if self.tables_zeroed_by_optimizer:
            return
        """per-frame tables: only row ``time_id`` gets a gradient, every other row must read zero.  With a
        FlatGradBuffer the four tables are adjacent: one fill."""
        m = self.model
        tabs = [t.grad for t in (m.sk_r, m.sk_d_rot, m.sk_d_scale, m.global_tr) if t is not None]
        lo = min(t.data_ptr() for t in tabs)
        total = sum(t.numel() for t in tabs)
        first = min(tabs, key=lambda t: t.data_ptr())
        contiguous = (max(t.data_ptr() + t.numel() * 4 for t in tabs) - lo == total * 4
                      and first.untyped_storage().data_ptr() == tabs[-1].untyped_storage().data_ptr())
        if contiguous:
            torch.as_strided(first, (total,), (1,)).zero_()
        else:
            for t in tabs:
                t.zero_()

    def _raster_inputs(self, rs: Optional[GaussianRasterizationSettings]) -> '_C._RasterInputs':
        m = self.model
        a = _C._RasterInputs()
        vt = self.view_table if rs is None else None
        if vt is not None:
            rs = vt.settings  # the launch arguments common to all views; the camera itself is read from the slot
        a.P, a.sh_degree, a.sh_coeffs, a.E = self.P, int(rs.sh_degree), (m.max_sh_degree + 1) ** 2, 0
        a.image_height, a.image_width = self.H, self.W
        a.tanfovx, a.tanfovy, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
        a.prefiltered, a.debug, a.colmap = int(bool(rs.prefiltered)), int(bool(rs.debug)), int(bool(rs.colmap))
        if vt is not None:
            from sk_gs_amd import view_slot as vsl
            a.viewmatrix, a.projmatrix, a.campos = vt.ptr(vsl.W_VIEW), vt.ptr(vsl.W_PROJ), vt.ptr(vsl.W_CAMPOS)
            a.tanfov_device = vt.ptr(vsl.W_TANFOV)
        else:
            for name in ('viewmatrix', 'projmatrix', 'campos'):
                t = getattr(rs, name)
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), name
                setattr(a, name, t.data_ptr())
        a.means3D, a.opacity = self.means.data_ptr(), self.opacity.data_ptr()
        a.scales, a.rotations = self.scales.data_ptr(), self.rotations.data_ptr()
        a.sh, a.sh_rest = m._features_dc.data_ptr(), m._features_rest.data_ptr()
        a.background = None if self.background is None else self.background.data_ptr()
        a.tile_bucket_capacity = self.tile_bucket
        a.tiles_per_gaussian_hint = int(getattr(self, 'tiles_per_gaussian_hint', 0))
        a.live_count = None if self._live is None else self._live.data_ptr()
        return a

    def _deform_inputs(self, time_id: int) -> '_C._DeformInputs':
        m = self.model
        a = _C._DeformInputs()
        a.P, a.K, a.M = self.P, self.K, self.M
        a.points = a.xyz = m._xyz.data_ptr()  # points = xyz.detach() (sk_gs.py:1113): same storage
        a.weights, a.indices, a.bone_T = self.weights.data_ptr(), self.indices.data_ptr(), self.bone_T.data_ptr()
        if self.deform_net is None:
            assert time_id is not None
            a.bone_drot, a.bone_dscale = m.sk_d_rot[time_id].data_ptr(), m.sk_d_scale[time_id].data_ptr()
        else:
            a.bone_drot, a.bone_dscale = self._d_rot.data_ptr(), self._d_scale.data_ptr()
        a.log_scale, a.rot, a.opacity_logit = m._scaling.data_ptr(), m._rotation.data_ptr(), m._opacity.data_ptr()
        a.live_count = None if self._live is None else self._live.data_ptr()
        return a

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _frame(self, time_id: Optional[int]):
        """(global_T pointer, its gradient's pointer, device frame-index pointer or None) of the frame: a row of the tables,
        or the tables' base + the slot's frame index"""
        m = self.model
        if time_id is None:
            from sk_gs_amd import view_slot as vsl
            # (a forward may run while the gradients are detached -- zero_grad(set_to_none=True) of a foreign loop: the gradient
            # pointers are only read by the backward launches)
            gg = m.global_tr.grad
            return m.global_tr.data_ptr(), (None if gg is None else gg.data_ptr()), C.c_void_p(self.view_table.ptr(vsl.W_FRAME))
        gg = m.global_tr.grad
        return m.global_tr[time_id].data_ptr(), (None if gg is None else gg[time_id].data_ptr()), None

    def forward(self, rs: Optional[GaussianRasterizationSettings] = None, time_id: Optional[int] = None):
        """bone chain -> KNN + LBS weights -> skin + activations -> rasterize.  Fills ``image`` / ``out_opacity``.
        Without arguments: the view selected in ``view_table``."""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        t = self._topo
        P, M, K = self.P, self.M, self.K
        assert (rs is None) == (time_id is None) and (rs is not None or self.view_table is not None)
        sk_r_raw = self._joint_rotations(time_id)  # (fused network: the kinematic chain ran in the same launch)
        gT, _, fidx = self._frame(time_id)
        # (running the single-workgroup bone-chain kernels on a forked stream beside the wide kernels was measured:
        # the fork/join edges of the captured graph cost more (+12 us per step) than the ~10 us of overlap)
        if self._mlp_fused is None:
            chk(lib.skgs_bone_chain_forward(
                C.c_int32(M), C.c_int32(t['root']), _p(t['parents']), _p(t['level_nodes']), _p(t['level_start']),
                C.c_int32(t['num_levels']), _p(sk_r_raw), _p(m.joints), C.c_void_p(gT), _p(self.bone_T),
                _p(self.chain_A), fidx, st))
        d = self._deform_inputs(time_id)
        assert not (self.wide and self._live is not None), 'row capacity: the one-launch skinning path (M <= 60, K <= 8)'
        if self.lbs_method != 'W':  # K nearest bones + kernel / dist weighting (raw radius / weight parameters: the kernel
            chk(lib.skgs_knn_dist_weights_forward(  # applies exp / sigmoid itself), then the skinning
                C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(3), C.c_void_p(d.points), _p(m.joints),
                _p(m._sp_radius), _p(m._sp_weight), C.c_float(m.lbs_temperature), C.c_int32(1), _p(self.indices),
                _p(self.weights), _p(self.nn_dist), st))
            chk(lib.skgs_lbs_deform_forward(C.byref(d), _p(self.means), _p(self.scales), _p(self.rotations), _p(self.opacity),
                                            None, None, None, st))
            a = self._raster_inputs(rs)
            chk(lib.skgs_rasterize_forward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.image),
                                           _p(self.out_opacity), None, None, st))
            return a, d
        if self.wide:  # many bones: search + softmax, then the skinning, as two launches (bone tables stay in global memory)
            chk(lib.skgs_knn_lbs_weights(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_void_p(d.points), _p(m.joints),
                                         _p(m.sp_W), _p(self.indices), _p(self.weights), st))
            chk(lib.skgs_lbs_deform_forward(C.byref(d), _p(self.means), _p(self.scales), _p(self.rotations), _p(self.opacity),
                                            None, None, None, st))
            a = self._raster_inputs(rs)
            chk(lib.skgs_rasterize_forward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.image),
                                           _p(self.out_opacity), None, None, st))
            return a, d
        if self.deform_in_preprocess and K <= 8:
            # K nearest bones + softmax weights + skinning + activations as a job of the rasterizer's per-Gaussian launch: the
            # lane that projects a Gaussian computes its mean / scale / rotation / opacity first (written for the backward)
            j = _C._KnnDeformJob()
            j.M, j.K = M, K
            j.points, j.joints, j.sp_W, j.bone_T = d.points, m.joints.data_ptr(), m.sp_W.data_ptr(), d.bone_T
            j.bone_drot, j.bone_dscale, j.xyz, j.log_scale = d.bone_drot, d.bone_dscale, d.xyz, d.log_scale
            j.rot, j.opacity_logit, j.out_idx, j.out_weights = d.rot, d.opacity_logit, self.indices.data_ptr(), self.weights.data_ptr()
            j.means, j.scales = self.means.data_ptr(), self.scales.data_ptr()
            j.rotations, j.opacity = self.rotations.data_ptr(), self.opacity.data_ptr()
            a = self._raster_inputs(rs)
            a.deform_job = C.pointer(j)
            chk(lib.skgs_rasterize_forward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.image),
                                           _p(self.out_opacity), None, None, st))
            a.deform_job = None  # (the backward's copy of the inputs: the job was the forward's)
            return a, d
        # K nearest bones + softmax weights + skinning + activations: one launch (weights / indices kept for the backward)
        chk(lib.skgs_knn_lbs_deform_forward(
            C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_void_p(d.points), _p(m.joints), _p(m.sp_W), C.c_void_p(d.bone_T),
            C.c_void_p(d.bone_drot), C.c_void_p(d.bone_dscale), C.c_void_p(d.xyz), C.c_void_p(d.log_scale), C.c_void_p(d.rot),
            C.c_void_p(d.opacity_logit), _p(self.indices), _p(self.weights), _p(self.means), _p(self.scales),
            _p(self.rotations), _p(self.opacity), _p(self._live), st))
        a = self._raster_inputs(rs)
        chk(lib.skgs_rasterize_forward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.image),
                                       _p(self.out_opacity), None, None, st))
        return a, d

    @torch.no_grad()
    def forward_backward(self, rs: Optional[GaussianRasterizationSettings] = None, time_id: Optional[int] = None,
                         target: Optional[Tensor] = None):
        """one training view: (rs, time_id, target), or no arguments = the view selected in ``view_table``"""
        self.backward_raster(rs, time_id, target)
        self.backward_skinning(time_id)

    @torch.no_grad()
    def backward_raster(self, rs: Optional[GaussianRasterizationSettings] = None, time_id: Optional[int] = None,
                        target: Optional[Tensor] = None):
        """first half of the step: forward, loss, rasterizer backward.  On return the SH gradients
        (``_features_dc.grad``, ``_features_rest.grad``) are final; the skinning backward has not run yet."""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        P, M, K, W, H = self.P, self.M, self.K, self.W, self.H
        if rs is None:  # the target is an image of the table's stack, chosen by the slot
            from sk_gs_amd import view_slot as vsl
            assert target is None and self.view_table.targets is not None
            target, gt_index = self.view_table.targets, C.c_void_p(self.view_table.ptr(vsl.W_TARGET))
        else:
            gt_index = None
        assert target.is_cuda and target.dtype == torch.float32 and target.is_contiguous()
        self._zero_table_grads()
        a, d = self.forward(rs, time_id)
        # ---- loss and dL/dimage
        chk(lib.skgs_image_loss_forward(C.c_int32(3), C.c_int32(H), C.c_int32(W), _p(self.image), _p(target), gt_index,
                                        C.c_float(self.lambda_l1), C.c_float(self.lambda_ssim), None,  # value: backward
                                        _p(self.loss_ws), C.c_size_t(self.loss_ws.numel()), st))
        self._loss_backward_and_raster(a, d, target, gt_index, time_id, self.grad_scale)

    def _loss_backward_and_raster(self, a, d, target, gt_index, time_id, grad_loss: Tensor):
        """dL/dimage from the loss workspace (seeded with the device scalar ``grad_loss``), then the rasterizer backward"""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        W, H = self.W, self.H
        chk(lib.skgs_image_loss_backward(C.c_int32(3), C.c_int32(H), C.c_int32(W), _p(self.image), _p(target), gt_index,
                                         C.c_float(self.lambda_l1), C.c_float(self.lambda_ssim), _p(grad_loss),
                                         _p(self.loss_ws), C.c_size_t(self.loss_ws.numel()), _p(self.dL_dimage),
                                         _p(self.loss3), st))
        self._raster_backward(a, d, time_id)

    def _raster_backward(self, a, d, time_id):
        """the rasterizer backward on the cotangent in ``dL_dimage`` (written by the loss backward, or by a caller that holds
        d objective / d image itself: sk_gs_amd/reference_fused.py)"""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        # ---- rasterize backward: SH gradients land in the parameters' .grad, the rest feeds the skinning backward
        g = _C._RasterGrads()
        g.dL_dout_color = self.dL_dimage.data_ptr()  # dL_dout_opacity = NULL: the background term is in-kernel
        g.dL_dmeans2D, g.dL_dcolors, g.dL_dopacity = (self.grad_means2D.data_ptr(), self.g_colors.data_ptr(),
                                                      self.g_opacity.data_ptr())
        g.dL_dmeans3D, g.dL_dcov3D = self.g_means.data_ptr(), self.g_cov3D.data_ptr()
        if self.sh_factors is None:
            g.dL_dsh, g.dL_dsh_rest = m._features_dc.grad.data_ptr(), m._features_rest.grad.data_ptr()
        else:  # view-parallel: (direction, colour gradient) per Gaussian now, the SH rows after the all-gather
            g.dL_dsh_factors = self.sh_factors.data_ptr()
        g.dL_dscales, g.dL_drotations = self.g_scales.data_ptr(), self.g_rotations.data_ptr()
        g.workspace, g.workspace_bytes = self.bwd_ws.data_ptr(), self.bwd_ws.numel()
        g.workspace_is_zero = 1
        if self.densify_stats:  # this view's statistics, by the launch that writes the screen-space gradient
            g.stat_xyz_gradient_accum, g.stat_denom = self._acc_store.data_ptr(), self._den_store.data_ptr()
            g.stat_max_radii2D, g.stat_grad_multiplier = self._rad_store.data_ptr(), 1.0 / self._grad_scale_value
        self._rows_backward_done = False
        job = self._attach_backward_job(g, d, time_id)  # (kept alive until the call has read it)
        if job is not None and self._rows_backward_done and not self.keep_raster_grads:
            # the job takes the per-Gaussian gradients over in registers: their arrays are not written (80 B per Gaussian)
            g.dL_dcolors = g.dL_dopacity = g.dL_dmeans3D = g.dL_dcov3D = g.dL_dscales = g.dL_drotations = None
        chk(lib.skgs_rasterize_backward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.out_opacity),
                                        C.byref(g), st))

    def _attach_backward_job(self, g, d, time_id):
        """the skinning backward as a job of the rasterizer's per-Gaussian backward launch: fills the job pointer of ``g`` and
        sets ``_rows_backward_done`` (``backward_skinning`` then skips its launch); returns the struct to keep alive"""
        m = self.model
        if not (self.deform_backward_in_preprocess and self.lbs_method == 'W' and not self.wide and self.M <= 64 and self.K <= 8):
            return None
        # the skinning backward (with the softmax backward of the LBS logits) rides on the per-Gaussian launch: what
        # backward_skinning would launch next, on the same values, handed over in registers
        if self.deform_net is None:
            g_drot, g_dscale = m.sk_d_rot.grad[time_id], m.sk_d_scale.grad[time_id]
        else:
            _, g_drot, g_dscale = self._g_heads
        dense = self.spw_logit_grad is None
        j = _C._DeformBackwardJob()
        j.in_ = C.pointer(d)
        j.g_bone_T, j.g_bone_drot, j.g_bone_dscale = self.g_bone_T.data_ptr(), g_drot.data_ptr(), g_dscale.data_ptr()
        j.g_xyz, j.g_log_scale = m._xyz.grad.data_ptr(), m._scaling.grad.data_ptr()
        j.g_rot, j.g_opacity_logit = m._rotation.grad.data_ptr(), m._opacity.grad.data_ptr()
        j.g_sp_W = m.sp_W.grad.data_ptr() if dense else None
        j.g_logits = None if dense else self.spw_logit_grad.data_ptr()
        j.workspace, j.workspace_bytes = self.deform_ws.data_ptr(), self.deform_ws.numel()
        g.deform_backward_job = C.cast(C.pointer(j), C.c_void_p)
        self._rows_backward_done = True
        return j

    # ---- the same launches behind torch's autograd API ------------------------------------------------------------------
    @torch.no_grad()
    def forward_loss(self, rs: Optional[GaussianRasterizationSettings] = None, time_id: Optional[int] = None,
                     target: Optional[Tensor] = None) -> Tensor:
        """the forward half of a training view alone: deform, rasterize, ``0.8 L1 + 0.2 (1 - SSIM)`` -- ``loss3`` holds {total, L1,
        SSIM}; ``backward_pending`` runs the backward half later.  What ``loss()`` calls from an autograd node."""
        lib, st, chk = self.lib, _C._stream(), _C._check
        if rs is None:
            from sk_gs_amd import view_slot as vsl
            assert target is None and self.view_table.targets is not None
            target, gt_index = self.view_table.targets, C.c_void_p(self.view_table.ptr(vsl.W_TARGET))
        else:
            gt_index = None
        assert target.is_cuda and target.dtype == torch.float32 and target.is_contiguous()
        self._zero_table_grads()
        a, d = self.forward(rs, time_id)
        chk(lib.skgs_image_loss_forward(C.c_int32(3), C.c_int32(self.H), C.c_int32(self.W), _p(self.image), _p(target), gt_index,
                                        C.c_float(self.lambda_l1), C.c_float(self.lambda_ssim), _p(self.loss3),
                                        _p(self.loss_ws), C.c_size_t(self.loss_ws.numel()), st))
        self._pending = (a, d, target, gt_index, time_id)
        return self.loss3[0]

    @torch.no_grad()
    def backward_pending(self, grad_loss: Optional[Tensor] = None):
        """the backward half of the view ``forward_loss`` ran: every parameter's gradient is WRITTEN into its ``.grad`` storage
        (the fused kernels overwrite, they do not accumulate).  ``grad_loss``: d(objective) / d(loss) as a device scalar tensor
        (multiplied by the step's own ``grad_scale``)."""
        a, d, target, gt_index, time_id = self._pending
        self._pending = None
        seed = self.grad_scale  # (None: 1)
        if grad_loss is not None:
            seed = grad_loss.detach().reshape(1).to(torch.float32)
            if self.grad_scale is not None:
                seed = seed * self.grad_scale
            seed = seed.contiguous()
        self._loss_backward_and_raster(a, d, target, gt_index, time_id, seed)
        self.backward_skinning(time_id)

    def loss(self, rs: Optional[GaussianRasterizationSettings] = None, time_id: Optional[int] = None,
             target: Optional[Tensor] = None) -> Tensor:
        """One training view as an AUTOGRAD node over the fused launches: ``loss = step.loss(rs, time_id, target);
        loss.backward(); optimizer.step()`` -- the reference's loop shape (train.py:179-250) with this package's 11 launches per
        view instead of the operator path's ~30.  The forward half runs now, the backward half when autograd reaches the node
        (seeded with the incoming d/dloss, so ``(2 * loss).backward()`` or a sum with other terms behave); the parameters'
        gradients are WRITTEN into their existing ``.grad`` tensors -- call ``optimizer.zero_grad()`` (which keeps the tensors)
        before the forward, not between forward and backward, and do not expect accumulation over several views."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        return _FusedViewLoss.apply(self, rs, time_id, target, *params)

    @torch.no_grad()
    def sh_grads_from_factors(self, all_factors: Tensor, sh_degree: int):
        """``all_factors`` [n_views, P, 6] (this rank's ``sh_factors`` and everybody else's, in rank order) ->
        ``_features_dc.grad`` / ``_features_rest.grad``, the views summed in index order"""
        m = self.model
        n_views = all_factors.numel() // (self.P * 6)
        assert all_factors.is_contiguous() and all_factors.numel() == n_views * self.P * 6
        _C._check(self.lib.skgs_sh_grad_from_factors(
            C.c_int32(self.P), C.c_int32(n_views), C.c_int32(sh_degree), C.c_int32(1 + m._features_rest.shape[1]),
            _p(all_factors), _p(m._features_dc.grad), _p(m._features_rest.grad), _C._stream()))

    @torch.no_grad()
    def backward_skinning(self, time_id: Optional[int] = None, part: Optional[str] = None):
        """second half: skinning backward (Gaussian parameters' gradients written in place), LBS logits, bone chain.
        With ``spw_logit_grad`` set, the logit gradient is left compact ([P,K], for the all-reduce) and
        ``scatter_spw_grad`` expands it into ``sp_W.grad`` later.  ``part`` (fused network, ``lbs_method='W'``, K <= the fused
        limit): ``'rows'`` runs only the per-Gaussian launch -- on return the gradients of xyz / scaling / rotation / opacity and
        the LBS logits are final, a view-parallel step can put them on the wire -- and ``'skeleton'`` the rest (network,
        kinematic chain, joints, per-frame tables)."""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        assert part in (None, 'rows', 'skeleton')
        if part is not None:
            assert self.lbs_method == 'W' and not self.wide and self._mlp_fused is not None, \
                'backward_skinning(part=...): the one-launch skinning backward and the fused skeleton stage'
        if part == 'skeleton':
            self._time_id = time_id
            self._deform_net_backward()
            return
        t = self._topo
        P, M, K = self.P, self.M, self.K
        d = self._deform_inputs(time_id)
        self._time_id = time_id
        gT, g_gT, fidx = self._frame(time_id)
        if self.deform_net is None:
            sk_r_raw, g_raw = m.sk_r[time_id], m.sk_r.grad[time_id]
            g_drot, g_dscale = m.sk_d_rot.grad[time_id], m.sk_d_scale.grad[time_id]
        else:  # the three heads of the producer network: their gradients feed its backward below
            sk_r_raw, (g_raw, g_drot, g_dscale) = self._sk_r_raw, self._g_heads
        if self.lbs_method != 'W':
            chk(lib.skgs_lbs_deform_backward(
                C.byref(d), _p(self.g_means), _p(self.g_scales), _p(self.g_rotations), _p(self.g_opacity),
                _p(self.g_weights), _p(self.g_bone_T), _p(g_drot), _p(g_dscale),
                _p(m._xyz.grad), _p(m._scaling.grad), _p(m._rotation.grad), _p(m._opacity.grad), _p(self.deform_ws),
                C.c_size_t(self.deform_ws.numel()), st))
        elif self._rows_backward_done:  # (ran as a job of the rasterizer's backward, forward_backward above)
            # (the flag stays until the next backward_raster clears it: a GraphedSteps capture calls this function twice -- warm-up,
            # then the recording -- behind ONE backward_raster; consumed by the first call, the recording launched the rows pass a second
            # time, harmless only while the rasterizer still wrote the arrays that launch reads)
            if part == 'rows':
                return
        elif not self.wide:
            # skinning backward with the softmax backward of the LBS logits folded in: dense rows straight into sp_W.grad,
            # or the compact [P,K] gradient for the all-reduce
            dense = self.spw_logit_grad is None
            chk(lib.skgs_lbs_deform_backward_logits(
                C.byref(d), _p(self.g_means), _p(self.g_scales), _p(self.g_rotations), _p(self.g_opacity),
                None, _p(self.g_bone_T), _p(g_drot), _p(g_dscale),
                _p(m._xyz.grad), _p(m._scaling.grad), _p(m._rotation.grad), _p(m._opacity.grad),
                _p(m.sp_W.grad) if dense else None, None if dense else _p(self.spw_logit_grad), _p(self.deform_ws),
                C.c_size_t(self.deform_ws.numel()), st))
            if part == 'rows':
                return
        else:
            chk(lib.skgs_lbs_deform_backward(
                C.byref(d), _p(self.g_means), _p(self.g_scales), _p(self.g_rotations), _p(self.g_opacity),
                _p(self.g_weights), _p(self.g_bone_T), _p(g_drot), _p(g_dscale),
                _p(m._xyz.grad), _p(m._scaling.grad), _p(m._rotation.grad), _p(m._opacity.grad), _p(self.deform_ws),
                C.c_size_t(self.deform_ws.numel()), st))
            self._lbs_logits_backward()
        if self._mlp_fused is None:  # (fused network: the chain's backward runs inside its backward launch)
            chk(lib.skgs_bone_chain_backward(
                C.c_int32(M), C.c_int32(t['root']), _p(t['parents']), _p(t['level_nodes']), _p(t['level_start']),
                C.c_int32(t['num_levels']), _p(sk_r_raw), _p(m.joints), C.c_void_p(gT), _p(self.chain_A),
                _p(self.g_bone_T), _p(g_raw), _p(m.joints.grad) if getattr(m, 'learn_joints', False) else None,
                C.c_void_p(g_gT), fidx, st))
        if self.deform_net is not None:
            self._deform_net_backward()
        if self.lbs_method != 'W':
            # weights -> distances -> joints / radii / kernel weights.  AFTER the chain backward, which WRITES joints.grad:
            # this adds to it (the optimizer's closing launch adds the network-input part the same way)
            learn = getattr(m, 'learn_joints', False)
            chk(lib.skgs_knn_dist_weights_backward(
                C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(3), C.c_void_p(d.points), _p(m.joints), _p(m._sp_radius),
                _p(m._sp_weight), C.c_float(m.lbs_temperature), C.c_int32(1), C.c_int32(1), _p(self.weights), _p(self.indices),
                _p(self.nn_dist), _p(self.g_weights), None, _p(m.joints.grad) if learn else None,
                None if m._sp_radius is None else _p(m._sp_radius.grad), None if m._sp_weight is None else _p(m._sp_weight.grad),
                _p(self.dist_ws), C.c_size_t(self.dist_ws.numel()), st))
        # (densify_stats: folded into the rasterizer backward's per-Gaussian launch, see backward_raster)

    def _lbs_logits_backward(self):
        """softmax backward of the LBS logits as its own launch (many bones / neighbours: M > 64 or K > 8)"""
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        P, M, K = self.P, self.M, self.K
        if self.spw_logit_grad is None:
            chk(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(M), C.c_int32(K), _p(self.weights),
                                              _p(self.indices), _p(self.g_weights), _p(m.sp_W.grad), st))
        else:
            chk(lib.skgs_lbs_weights_backward_compact(C.c_int32(P), C.c_int32(K), _p(self.weights), _p(self.g_weights),
                                                      _p(self.spw_logit_grad), st))

    def _time_tensor(self, time_id: Optional[int]) -> Tensor:
        """the frame's time as a 1-element device tensor: a row of ``frame_times`` or the slot's time word"""
        if time_id is not None:
            return self.model.frame_times[time_id]
        from sk_gs_amd import view_slot as vsl
        return self.view_table.slot[vsl.W_TIME:vsl.W_TIME + 1]

    def input_grad_job(self):
        """the frequency-encoding backward that adds the network-input path to ``joints.grad``, as the argument tuple of
        ``FusedAdam.step_tail(freq_job=...)`` (None: joints are not trained)"""
        if self._g_x0 is None or self._mlp_fused is None:
            return None
        mlp = self.deform_net
        return (self.M, mlp.p_in, mlp.p_degree, self._g_x0, self._mlp_fused.x0, self._mlp_fused.x0.shape[1],
                self.model.joints.grad, True)

    def _bones_desc(self, time_id: Optional[int]):
        """the kinematic chain as a rider of the fused network launches (include/skgs.h::skgs_bone_chain_desc)"""
        from sk_gs_amd.deform_net import BoneChainDesc
        m, t = self.model, self._topo
        gT, g_gT, fidx = self._frame(time_id)
        b = BoneChainDesc()
        b.M, b.root, b.num_levels = self.M, t['root'], t['num_levels']
        b.parents, b.level_nodes, b.level_start = t['parents'].data_ptr(), t['level_nodes'].data_ptr(), t['level_start'].data_ptr()
        b.joints, b.global_T = m.joints.data_ptr(), gT
        b.frame_index = fidx.value if isinstance(fidx, C.c_void_p) else fidx
        b.bone_T, b.chain_A = self.bone_T.data_ptr(), self.chain_A.data_ptr()
        b.sk_r_raw, b.g_bone_T = self._sk_r_raw.data_ptr(), self.g_bone_T.data_ptr()
        b.g_joints = m.joints.grad.data_ptr() if (getattr(m, 'learn_joints', False) and m.joints.grad is not None) else None
        b.g_global_T = g_gT
        # every training step refreshes the frame's row of the test-time cache (sk_gs.py:1077-1079); written by the launch
        # (the frame's row: by pointer for an explicit time_id, base + the slot's frame index otherwise -- like global_T)
        cache = getattr(m, 'sk_cache', None)
        if cache is not None and cache.is_cuda and cache.shape[1] == self.M:
            b.sk_cache = cache.data_ptr() if time_id is None else cache[time_id].data_ptr()
        else:
            b.sk_cache = None
        return b

    def _joint_rotations(self, time_id: Optional[int]) -> Tensor:
        """raw joint rotations of the frame: a row of the per-frame table, or the producer network's first head (the
        network also fills d_rot / d_scale): 2 encode + 8 layer + 3 head launches (csrc/mlp.hip)"""
        m = self.model
        if self.deform_net is None:
            return m.sk_r[time_id]
        tt = self._time_tensor(time_id)
        if self._mlp_fused is not None:  # the whole skeleton stage -- network and kinematic chain -- in one launch
            if not self.skeleton_ahead:
                self.skeleton_forward(time_id)
            return self._sk_r_raw
        from sk_gs_amd.deform_net import _lin_fwd
        net, run = self.deform_net.dynamic_net, self._mlp
        run.encode(m.joints, tt, self._x0)
        run.forward_hidden(self._x0, self._acts)
        M, H, IN = self.M, net.dim_hidden, net.in_channels
        in1, in2 = net.layer_dims()[-1]
        o = 0
        for buf, oc in zip((self._sk_r_raw, self._d_rot, self._d_scale), net.out_channels):
            _lin_fwd(self.lib, M, in1, in2, oc, self._acts[-1].data_ptr(), H, self._x0.data_ptr() if in2 else None, IN,
                     net.last_weight[o:].data_ptr(), net.last_bias[o:].data_ptr(), buf.data_ptr(), oc, 0)
            o += oc
        return self._sk_r_raw

    def skeleton_forward(self, time_id: Optional[int] = None, side_adam=None):
        """the skeleton stage of a view (network heads + kinematic chain, one launch) on its own; ``side_adam``
        (``FusedAdam.side_range``): an optimizer piece for the CUs the launch leaves idle"""
        assert self._mlp_fused is not None
        self._mlp_fused.forward(self.model.joints, self._time_tensor(time_id),
                                head_out=(self._sk_r_raw, self._d_rot, self._d_scale), bones=self._bones_desc(time_id),
                                side_adam=side_adam)

    def _deform_net_backward(self):
        """weight gradients of the producer network, written into the parameters' .grad: one launch (fused kernels) or
        3 head + 8 layer launches"""
        if self._mlp_fused is not None:
            net = self.deform_net.dynamic_net
            grads = [g for l in net.net for g in (l.weight.grad, l.bias.grad)] + [net.last_weight.grad, net.last_bias.grad]
            side = None
            if self.side_optimizer is not None:  # the per-Gaussian rows' Adam update on the CUs this launch leaves idle
                side = self.side_optimizer[0].side_range(*self.side_optimizer[1:])
            self._mlp_fused.backward(self.model.joints, self._time_tensor(self._time_id), self._g_heads, grads, self._g_x0,
                                     side_adam=side, bones=self._bones_desc(self._time_id))
            # joints.grad (written by the bone-chain backward) += the network-input path; with ``defer_input_grad`` the
            # optimizer's closing launch does it (``input_grad_job``)
            if self._g_x0 is not None and not self.defer_input_grad:
                self._mlp.input_grad(self._g_x0, self._mlp_fused.x0, self.model.joints.grad, accumulate=True)
            return
        from sk_gs_amd.deform_net import _lin_bwd
        net, run = self.deform_net.dynamic_net, self._mlp
        M, H, IN = self.M, net.dim_hidden, net.in_channels
        in1, in2 = net.layer_dims()[-1]
        o = 0
        for j, (g_head, oc) in enumerate(zip(self._g_heads, net.out_channels)):
            _lin_bwd(self.lib, M, in1, in2, oc, self._acts[-1].data_ptr(), H, self._x0.data_ptr() if in2 else None, IN,
                     net.last_weight[o:].data_ptr(), None, g_head.data_ptr(), oc, 0, net.last_weight.grad[o:].data_ptr(),
                     net.last_bias.grad[o:].data_ptr(), self._g_act[0].data_ptr(), H, None, IN, 1 if j > 0 else 0)
            o += oc
        grads = [g for l in net.net for g in (l.weight.grad, l.bias.grad)]
        run.backward_hidden(self._x0, self._acts, grads, self._g_act, self._g_x0)
        if self._g_x0 is not None:
            run.input_grad(self._g_x0, self._x0, self.model.joints.grad, accumulate=True)

    @torch.no_grad()
    def scatter_spw_grad(self):
        """expand the (all-reduced) compact logit gradient into the dense ``sp_W.grad``"""
        _C._check(self.lib.skgs_lbs_logits_scatter(C.c_int32(self.P), C.c_int32(self.M), C.c_int32(self.K),
                                                   _p(self.indices), _p(self.spw_logit_grad), _p(self.model.sp_W.grad),
                                                   _C._stream()))

    @torch.no_grad()
    def add_densification_stats(self):
        """accumulate this view's statistics (one launch); ``forward_backward`` calls it when ``densify_stats`` is set"""
        # the backward may be pre-scaled (grad_scale = 1 / world): the statistic is the norm of the UNSCALED screen-space
        # gradient (gaussian_splatting.py:503-513), or N ranks would densify as if max_grad were N times larger
        # (rows beyond the live count carry radius 0 -- skgs_raster_inputs.live_count -- and are skipped like culled ones)
        _C._check(self.lib.skgs_densify_stats(C.c_int32(self.P), _p(self.radii), _p(self.grad_means2D),
                                             C.c_float(1.0 / self._grad_scale_value), _p(self._acc_store),
                                             _p(self._den_store), _p(self._rad_store), _C._stream()))

    @torch.no_grad()
    def grow_capacity(self, factor: float = 2.0):
        """re-allocate the binning buffer with ``factor`` times the tile-instance capacity (or slots per tile bucket).  Every
        graph captured with the old buffer is invalid afterwards: re-capture.  The sticky overflow counter restarts at 0."""
        lib = self.lib
        if self.tile_bucket > 0:
            self.tile_bucket = ((int(self.tile_bucket * factor) + 63) // 64) * 64
            capacity = ((self.W + 15) // 16) * ((self.H + 15) // 16) * self.tile_bucket
        else:
            capacity = int(int(lib.skgs_binning_capacity(C.c_size_t(self.binning.numel()))) * factor) + 1024
        self.binning = torch.empty((lib.skgs_binning_buffer_bytes(C.c_int64(int(capacity))),), dtype=torch.uint8,
                                   device=self.binning.device)
        self.geom[:256].zero_()
        self._bufs = _C._buffers(self.geom, self.binning, self.img)
        return capacity

    def status(self) -> dict:
        """(synchronising) num_rendered / overflow / longest tile list of the last forward, and the number of
        forwards since construction whose tile lists did not fit ``capacity`` (``overflow_events``); with the fused
        deform network also ``mlp_failed`` (launches whose in-kernel exchange gave up: must be 0)"""
        st = _C.read_status(self.geom)
        if self.deform_net is not None and self._mlp_fused is not None:
            m = self._mlp_fused.status()
            st['mlp_failed'] = m['failed']
            # launches whose network sat on ONE XCD by their own census, the exchange kept in its L2 (skgs_deform_mlp_xcd_mode)
            st['mlp_one_xcd'] = dict(forward=[m['one_xcd_forward'], m['forward']], backward=[m['one_xcd_backward'], m['backward']],
                                     xcds=[m['xcds_forward'], m['xcds_backward']])
        return st


class _FusedViewLoss(torch.autograd.Function):
    """``FusedViewStep.loss``: forward = ``forward_loss``, backward = ``backward_pending`` (side effect: the parameters' ``.grad``
    storage is overwritten; the node itself hands autograd no gradient tensors -- returning them would make autograd add them on
    top of what the kernels wrote)."""

    @staticmethod
    def forward(ctx, step, rs, time_id, target, *params):
        ctx.step = step
        return step.forward_loss(rs, time_id, target).clone()  # (a copy: `loss3` is rewritten by the next view)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ctx.step.backward_pending(g)
        arm = getattr(ctx.step, '_arm_tail', None)
        if arm is not None:     # FusedTrainStep.loss(): the next optimizer.step() is THIS step's closing launch
            ctx.step._arm_tail = None
            arm[0]._pending_tail = arm[1]
        return (None,) * (4 + len([p for p in ctx.step.model.parameters() if p.requires_grad]))
