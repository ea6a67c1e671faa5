"""The fused per-view step UNDER the unmodified reference: ``SkeletonGaussianSplatting.render`` + the ``rgb`` / ``ssim`` loss terms.

``accelerate_reference()`` (sk_gs_amd/reference_accel.py) gives leaf methods of the reference a fast path, but the iteration stays
the reference's eager Python: ``forward`` (networks/sk_gs.py:1160-1204: ~130 launches through the stand-ins), ``render`` (:1206-1242:
the rasterizer adapter + ten ``torch.stack`` copies), the image terms of ``loss`` (:1524-1529).  This module puts the SAME launches the
package's own trainer issues (``fused_step.FusedViewStep``: skeleton forward | preprocess forward with the skinning as its job |
scatter | sort | blend forward || loss forward || loss backward | blend backward | preprocess backward with the skinning backward |
bone-moment finalize | skeleton backward) behind those two methods, on the model's OWN ``nn.Parameter`` objects:

* ``render(self, *args, t, info, background, time_id, scale_modifier, stage, **kwargs)`` -- when the conditions below hold -- fills the
  live view slot from the DEVICE tensors of ``info`` (``skgs_view_slot_fill``: the reference's ``prepare_inputs`` reads them back to
  the host with ``math.tan(0.5 * FoV[b, 0])``, a blocking copy per iteration), runs the five forward launches and returns the dict
  ``loss()`` / ``adaptive_control()`` / ``train_step`` read: ``images`` [1,H,W,3] (an autograd output of ONE node), ``stage``,
  ``radii``, ``viewspace_points`` (its ``.grad`` is filled by the backward), ``points``, ``_knn_w``, ``_knn_i``, ``_skT``, ``_sk_rot``,
  ``_sk_scale``; ``_d_xyz`` / ``_d_rot`` / ``_d_scale`` / ``visibility_filter`` are computed on first access.  The frame's row of
  ``sk_cache`` is written by the skeleton launch (:1077-1079).
* ``ImageLoss.forward`` (``method='l1'``, unmasked; networks/losses/image_loss.py:19-32) and ``SSIM_Loss.forward`` recognise an image
  that comes from that node and return the two terms as outputs of ONE second node (one loss-forward launch for both); its backward
  is one launch that hands ``d / d image`` to the render node in place, whose backward runs the six backward launches and WRITES the
  parameters' gradients into persistent ``.grad`` tensors (re-attached after the reference's ``zero_grad(set_to_none=True)``; a
  gradient somebody else already accumulated in the same pass is added, so further loss terms on the same parameters stay exact).
  Any other use of ``images`` (another loss, ``images * 2``, ...) reaches the render node as an ordinary cotangent: still exact, one
  copy slower.

Conditions of the fast path (anything else runs the reference's own ``render``; ``calls`` counts both and ``why_not`` keeps the last
reason): stage ``sk`` or ``sp``; training with grad enabled; ONE view; ``t`` / ``time_id`` / ``info`` tensors on the HIP device; no ``hook``
(other keywords are ignored, as the reference's ``render`` ignores them); ``use_official_gaussians_render`` (the shipped configs), no
``convert_SHs_python`` / ``compute_cov3D``; a background of <= 3 values.  Stage ``sk``: the skeleton is initialised; ``LBS_method == 'W'``
with ``sp_W`` [P, M] over the M <= 48 joints, ``num_knn`` <= 8; quaternion rotations, no ``sk_feature``; a ``SimpleDeformationNetwork`` the
one-launch kernels cover (width 256, frequency encoders) -- its three head matrices (``dynamic_net.last``) are RE-HOMED into one contiguous
matrix (their ``.data`` become row views of it: same Parameter objects, same values; the kernels read and train them in place).  Stage
``sp`` (``FusedSuperpointStep``): 60 < M <= 1024 superpoints, ``num_knn`` <= 8, 0 or 8 hyper dimensions, any of the four weightings,
``warp_method`` LBS / LBS_c / largest, ``sep_rot`` either way, ``is_blender=True``; ``outputs['_knn_w']`` and ``outputs['_spT']`` are outputs
of the node too -- the shipped ``sparse`` / ``smooth`` regularisers and the joint losses differentiate them (sk_gs.py:1555-1574) -- and their
cotangents enter the backward half (``skgs_sp_skinning_job.g_weights_extra``; ``g_bone_T`` before the network's backward).

The forward half and the backward half are ONE hipGraph replay each (``SKGS_REF_FUSED_GRAPHS=0``: the same launches issued one by one);
per call the host still issues the slot fill, the target's layout copy and the two loss launches.  Nothing blocks on the device per
iteration: every ``CHECK_EVERY`` (64) calls the route reads the status words once (a tile-list overflow doubles the bucket capacity with
a warning; the reference itself blocks on ``num_rendered`` in every forward), and at calls 1 / 32 / 128 / 512 / every 1024th it
re-measures the longest tile list with one extra compact-list forward and resizes the per-tile buckets to 1.5 x that (both directions).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import types
import warnings
import weakref

import torch

calls = {'render_fused': 0, 'render_reference': 0, 'image_terms_fused': 0, 'image_terms_reference': 0, 'backward_direct': 0,
         'backward_cotangent': 0, 'foreign_grads_added': 0, 'routes_built': 0, 'capacity_grown': 0, 'capacity_retuned': 0, 'backward_extras': 0}
why_not = {'render': None, 'terms': None}
_routes = weakref.WeakKeyDictionary()     # reference model -> FusedReferenceRoute | str (the reason there is none)
CHECK_EVERY = 64
_VIEW_NODES = ('ViewBackward', 'PermuteBackward', 'UnsqueezeBackward', 'SqueezeBackward', 'SliceBackward', 'AliasBackward',
               'UnsafeViewBackward', 'ReshapeAliasBackward', 'SelectBackward', 'TransposeBackward', 'ExpandBackward')


def _p(t):
    return C.c_void_p(None if t is None else t.data_ptr())


_frozen = False


def _freeze_collector_once():
    """``gc.freeze()`` when the first route is built (``SKGS_GC_FREEZE=0``: leave the collector alone).  An eager iteration creates a few
    hundred container objects (autograd nodes, ctypes structures, dicts), so CPython's oldest-generation collection comes round every
    ~230 iterations -- and walks EVERYTHING the process has imported: 103 ms with torch and the reference loaded (266 k tracked objects;
    tools/find_host_spikes.py), i.e. 0.45 ms per iteration against a 0.5 ms iteration.  Freezing moves what exists now (modules, classes,
    the model) into the permanent generation: later collections see only what training allocates (reference counting still frees
    everything it can; what is frozen is merely never scanned for cycles again)."""
    global _frozen
    if _frozen or os.environ.get('SKGS_GC_FREEZE', '1') == '0':
        return
    import gc
    gc.collect()
    gc.freeze()
    _frozen = True


class _LiveSlot:
    """the part of ``view_slot.ViewTable`` the step reads: ONE live record, filled per call from the reference's ``info``"""
    targets = None

    def __init__(self, dev, W, H, sh_degree, scale_modifier):
        from sk_gs_amd import view_slot
        self.slot = torch.zeros(view_slot.SLOT_WORDS, dtype=torch.float32, device=dev)
        self.settings = types.SimpleNamespace(image_height=int(H), image_width=int(W), tanfovx=1.0, tanfovy=1.0, sh_degree=int(sh_degree),
                                              scale_modifier=float(scale_modifier), prefiltered=False, debug=False, colmap=True)

    def ptr(self, word: int) -> int:
        return self.slot.data_ptr() + 4 * word

    def advance(self):
        return None


class _ModelView:
    """what ``FusedViewStep`` asks of a model, answered by the reference's ``SkeletonGaussianSplatting`` (same Parameter objects)"""
    static, capacity, lbs_method, lbs_temperature, learn_joints = False, None, 'W', 1.0, True
    sk_r = sk_d_rot = sk_d_scale = _sp_radius = _sp_weight = None

    def __init__(self, ref, shadow_net, topo):
        self._ref, self.sk_deform_net, self._topo = ref, shadow_net, topo
        for name in ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_W', 'joints', 'global_tr'):
            setattr(self, name, getattr(ref, name))
        self.P, self.M, self.K = int(ref._xyz.shape[0]), int(ref.joints.shape[0]), int(ref.num_knn)
        self.max_sh_degree = int(ref.max_sh_degree)
        self.sk_cache = ref.sk_cache

    def topology(self):
        return self._topo

    def parameters(self):
        net = self.sk_deform_net.dynamic_net
        return ([self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity, self.sp_W, self.joints,
                 self.global_tr] + [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias])


def _rehome_heads(ref_net, shadow):
    """the reference's head Linears (``dynamic_net.last``) become row views of the shadow's ONE head matrix / bias vector"""
    net, heads = shadow.dynamic_net, ref_net.dynamic_net.last
    if getattr(shadow, '_heads_rehomed', None) is not None and all(
            h.weight.data_ptr() == net.last_weight.data_ptr() + 4 * o * net.last_weight.shape[1] for h, o in zip(heads, shadow._heads_rehomed)):
        return
    with torch.no_grad():
        net.last_weight.data = torch.cat([h.weight.detach() for h in heads]).contiguous()
        net.last_bias.data = torch.cat([h.bias.detach() for h in heads]).contiguous()
        offs, o = [], 0
        for h in heads:
            oc = h.weight.shape[0]
            h.weight.data, h.bias.data = net.last_weight.data[o:o + oc], net.last_bias.data[o:o + oc]
            offs.append(o)
            o += oc
    net.last_weight.requires_grad_(True), net.last_bias.requires_grad_(True)
    shadow._heads_rehomed = offs


def unhome_heads(ref_net):
    """give the heads storage of their own again (``restore_reference``)"""
    from sk_gs_amd import reference_accel as ra
    sh = ra._shadows.get(ref_net)
    if sh is None or getattr(sh, '_heads_rehomed', None) is None:
        return
    with torch.no_grad():
        for h in ref_net.dynamic_net.last:
            h.weight.data, h.bias.data = h.weight.data.clone(), h.bias.data.clone()
    sh._heads_rehomed = None


def _conditions(ref):
    """None when ``ref`` (a SkeletonGaussianSplatting in stage sk) is what the fused step covers, else the reason"""
    from sk_gs_amd import _C, reference_accel as ra
    from sk_gs_amd.deform_net import fused_supported
    if not getattr(ref, 'use_official_gaussians_render', False):
        return 'use_official_gaussians_render is off (the in-tree rasterizer convention: operator path)'
    if getattr(ref, 'convert_SHs_python', False) or getattr(ref, 'compute_cov3D', False):
        return 'convert_SHs_python / compute_cov3D'
    if getattr(ref, 'LBS_method', None) != 'W' or ref.sp_W is None:
        return f'LBS_method {getattr(ref, "LBS_method", None)!r} (the fused route covers W)'
    if getattr(ref, 'sk_feature', None) is not None or getattr(ref, '_R_dim', 4) != 4:
        return 'sk_feature / lie rotations'
    ps = [ref._xyz, ref._features_dc, ref._features_rest, ref._scaling, ref._rotation, ref._opacity, ref.sp_W, ref.joints, ref.global_tr]
    if not all(torch.is_tensor(p) and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        return 'parameters are not contiguous fp32 tensors on a HIP device'
    P, M, K = ref._xyz.shape[0], ref.joints.shape[0], int(ref.num_knn)
    if P == 0 or tuple(ref.sp_W.shape) != (P, M) or not (1 <= K <= 8 and K <= M) or M > min(48, _C.fused_lbs_max_bones()):
        return f'shapes: P = {P}, sp_W {tuple(ref.sp_W.shape)}, {M} joints, K = {K}'
    if ref.joint_parents.dim() != 2 or ref.joint_parents.shape[0] != M or not ref.joint_parents.is_cuda:
        return 'joint_parents'
    if not bool(ref.sk_is_init):          # (one read-back when the route is built, not per step)
        return 'the skeleton is not initialised yet'
    if not (torch.is_tensor(ref.sk_cache) and ref.sk_cache.is_cuda and ref.sk_cache.dim() == 3 and ref.sk_cache.shape[1:] == (M, 11)
            and ref.sk_cache.is_contiguous() and ref.global_tr.shape[0] == ref.sk_cache.shape[0] and ref.global_tr.shape[1] == 7):
        return 'sk_cache / global_tr shapes'
    sh = ra.sk_net_shadow(ref.sk_deform_net)
    if sh is None or not fused_supported(sh, M) or tuple(sh.dynamic_net.out_channels) != (4, 4, 3):
        return 'sk_deform_net is not a network the one-launch kernels cover'
    return None


class _ModelViewSp:
    """what ``FusedSuperpointStep`` asks of a model (``superpoint.SuperpointGaussians``), answered by the reference's model in stage sp"""
    static, capacity, lbs_temperature, time_noise, sk_deform_net = False, None, 1.0, 0.0, None

    def __init__(self, ref, shadow_net):
        self._ref, self.sp_deform_net = ref, shadow_net
        for name in ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_points', 'hyper_feature',
                     'sp_hyper_feature', 'sp_W', '_sp_radius', '_sp_weight'):
            setattr(self, name, getattr(ref, name, None))
        self.P, self.M, self.K = int(ref._xyz.shape[0]), int(ref.sp_points.shape[0]), int(ref.num_knn)
        self.hyper_dim = int(ref.hyper_dim) if ref.hyper_feature is not None else 0
        self.max_sh_degree = int(ref.max_sh_degree)
        self.lbs_method, self.warp_method, self.sep_rot = ref.LBS_method, ref.warp_method, bool(ref.sep_rot)

    def topology(self):
        return {}

    def parameters(self):
        # (sp_points: detached everywhere on this path, sk_gs.py:753-755,845 -- except by the re-centring of LBS_c; a parameter the
        # reference gives no gradient must not get one here: Adam would move it on its old moments)
        ps = [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]
        if self.sp_W is not None:
            ps.append(self.sp_W)
        else:
            ps += [q for q in (self.hyper_feature, self.sp_hyper_feature, self._sp_radius, self._sp_weight) if q is not None]
        if self.warp_method == 'LBS_c':
            ps.append(self.sp_points)
        return ps + list(self.sp_deform_net.parameters())


def _conditions_sp(ref):
    """None when ``ref`` in stage sp is what ``FusedSuperpointStep`` covers, else the reason"""
    from sk_gs_amd import reference_accel as ra
    if not getattr(ref, 'use_official_gaussians_render', False):
        return 'use_official_gaussians_render is off (the in-tree rasterizer convention: operator path)'
    if getattr(ref, 'convert_SHs_python', False) or getattr(ref, 'compute_cov3D', False):
        return 'convert_SHs_python / compute_cov3D'
    if getattr(ref, 'warp_method', None) not in ('LBS', 'LBS_c', 'largest'):
        return f'warp_method {getattr(ref, "warp_method", None)!r}'
    if getattr(ref, 'LBS_method', None) not in ('W', 'dist', 'kernel', 'weighted_kernel'):
        return 'LBS_method'
    ps = [ref._xyz, ref._features_dc, ref._features_rest, ref._scaling, ref._rotation, ref._opacity, ref.sp_points] + \
         [q for q in (ref.hyper_feature, ref.sp_hyper_feature, ref.sp_W, ref._sp_radius, ref._sp_weight) if q is not None]
    if not all(torch.is_tensor(p) and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        return 'parameters are not contiguous fp32 tensors on a HIP device'
    P, M, K = ref._xyz.shape[0], ref.sp_points.shape[0], int(ref.num_knn)
    F_ = int(ref.hyper_dim) if ref.hyper_feature is not None else 0
    if P == 0 or not (1 <= K <= 8 and K <= M) or not (60 < M <= 1024) or F_ not in (0, 8):
        return f'shapes: P = {P}, {M} superpoints, K = {K}, {F_} hyper dimensions'
    if F_ and (tuple(ref.hyper_feature.shape) != (P, F_) or tuple(ref.sp_hyper_feature.shape) != (M, F_)):
        return 'hyper feature shapes'
    if ref.LBS_method == 'W' and (ref.sp_W is None or tuple(ref.sp_W.shape) != (P, M)):
        return 'sp_W shape'
    if ref.LBS_method in ('kernel', 'weighted_kernel') and (ref._sp_radius is None or ref._sp_radius.shape[0] != M):
        return '_sp_radius shape'
    if not getattr(ref.sp_deform_net, 'is_blender', False):
        return 'sp_deform_net without the time network (its per-call time noise, sk_gs.py:837-839, stays with the reference)'
    sh = ra.sp_net_shadow(ref.sp_deform_net)
    if sh is None or not sh.kernel_supported():
        return 'sp_deform_net is not a network the row-block kernels cover'
    return None


class FusedReferenceRoute:
    """one reference model's fused step for one stage: the adapter, the ``FusedViewStep`` / ``FusedSuperpointStep``, the persistent
    gradients, the graphs"""

    def __init__(self, ref, W, H, sh_degree, scale_modifier, stage='sk'):
        from sk_gs_amd import _C, reference_accel as ra
        self.lib = _C.load_library()
        self.ref = weakref.ref(ref)
        self.W, self.H, self.stage = int(W), int(H), stage
        dev = ref._xyz.device
        self.table = _LiveSlot(dev, W, H, sh_degree, scale_modifier)
        self._bucket = 0
        if stage == 'sk':
            shadow = ra.sk_net_shadow(ref.sk_deform_net)
            _rehome_heads(ref.sk_deform_net, shadow)
            self.shadow = shadow
            topo = ra._topology(ref.joint_parents, ref.joint_root)
            self.view = _ModelView(ref, shadow, topo)
        else:
            self.shadow = ra.sp_net_shadow(ref.sp_deform_net)
            self.view = _ModelViewSp(ref, self.shadow)
        saved = {p: p.grad for p in self.view.parameters()}
        stores = ()
        try:        # (the step's constructor builds zeroed gradient tensors for parameters that have none: those become the persistent ones)
            for p in saved:
                p.grad = None
            self.step = self._build_step(64)
            self.grads = {p: p.grad for p in self.view.parameters()}        # persistent: the kernels' write targets
            if stage == 'sk':
                net, heads = shadow.dynamic_net, ref.sk_deform_net.dynamic_net.last
                for h, o in zip(heads, shadow._heads_rehomed):               # the heads' gradients: row views of the head matrix's
                    oc = h.weight.shape[0]
                    self.grads[h.weight], self.grads[h.bias] = net.last_weight.grad[o:o + oc], net.last_bias.grad[o:o + oc]
                stores = (net.last_weight, net.last_bias)
        finally:    # whatever the caller's gradients were, they are back
            for p, g in saved.items():
                p.grad = g
            for p in stores:
                p.grad = None
        self._stores = stores
        net = types.SimpleNamespace(last_weight=stores[0], last_bias=stores[1]) if stores else types.SimpleNamespace(last_weight=None, last_bias=None)
        self.vp = torch.zeros((self.view.P, 3), dtype=torch.float32, device=dev, requires_grad=True)   # outputs['viewspace_points'][0]
        self.loss_ring = torch.zeros((64, 3), dtype=torch.float32, device=dev)
        self._ring_pos = 0
        self.target = torch.zeros((3, self.H, self.W), dtype=torch.float32, device=dev)    # the view's ground truth as [3,H,W]
        self.serial, self._fwd, self._terms, self._bg_key, self._spare = 0, None, None, None, None
        self._seen_events = 0
        self._attach = [(p, g, (p is net.last_weight or p is net.last_bias)) for p, g in self.grads.items()]
        self._tiles = None
        if stage == 'sp':   # cotangents the reference's loss puts on outputs['_knn_w'] / outputs['_spT'] (sparse, smooth, joint: sk_gs.py:1555-1574)
            self.gw_extra = torch.zeros((self.view.P, self.view.K), dtype=torch.float32, device=dev)
            self.gT_extra = torch.zeros((self.view.M, 7), dtype=torch.float32, device=dev)
            if self.view.sp_W is not None and os.environ.get('SKGS_REF_TILED_ADAM', '1') != '0':
                # LBS_method W (the shipped default): the dense [P, M] logit table receives a gradient at a row's K neighbours only.  The
                # patched torch.optim.Adam.step (reference_accel.adam_step) updates the 32-column tiles that have ever held one instead of
                # all P x M elements -- exact, see skgs_adam_masked_rows -- when the table carries this note; every backward says whether
                # anything besides this step's neighbours may have written to the gradient (then the update scans the gradient rows first)
                self._tiles = ra.LogitTiles(self.step.indices)
                self.view.sp_W._skgs_logit_tiles = self._tiles
        self.graphs, self._graph_key = None, None      # hipGraphs of the forward half / the backward half (None: eager launches)
        self.use_graphs = os.environ.get('SKGS_REF_FUSED_GRAPHS', '1') != '0'
        calls['routes_built'] += 1
        _freeze_collector_once()

    # -------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def identity(ref):
        net = ref.sk_deform_net.dynamic_net
        ps = [ref._xyz, ref._features_dc, ref._features_rest, ref._scaling, ref._rotation, ref._opacity, ref.sp_W, ref.joints,
              ref.global_tr] + list(net.net.parameters()) + list(net.last.parameters())
        return (tuple(id(p) for p in ps), tuple(p.data_ptr() for p in ps), int(ref._xyz.shape[0]), ref.joint_parents.data_ptr(),
                ref.joint_parents._version, ref.sk_cache.data_ptr(), int(ref.num_knn))

    def _build_step(self, bucket):
        from sk_gs_amd.fused_step import FusedViewStep
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        bg = torch.zeros(3, dtype=torch.float32, device=self.view._xyz.device)
        old = getattr(self, 'step', None)
        if old is not None:
            bg = old.background
        if self.stage == 'sp':
            from sk_gs_amd.superpoint import FusedSuperpointStep
            step = FusedSuperpointStep(self.view, self.W, self.H, capacity=max(24 * self.view.P, 64 * T) if bucket == 0 else 0,
                                       lambda_dssim=0.2, background=bg, tile_bucket=bucket, view_table=self.table)
        else:
            step = FusedViewStep(self.view, self.W, self.H, capacity=max(24 * self.view.P, 64 * T) if bucket == 0 else 0, lambda_dssim=0.2,
                                 background=bg, tile_bucket=bucket, view_table=self.table)
        self._bucket = bucket
        return step

    RETUNE_AT = (1, 32, 128, 512)      # calls at which the tile-list capacity is re-measured; afterwards every RETUNE_EVERY
    RETUNE_EVERY = 1024

    def _measure_longest(self):
        """one extra forward of the view in the slot with COMPACT tile lists (exact counts: the bucket layout's status words do not
        carry them) into a spare binning buffer, and one status read: (longest tile list, tile instances)"""
        from sk_gs_amd import _C
        st, lib = self.step, self.lib
        if getattr(self, '_spare', None) is None:
            T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
            self._spare = torch.empty((lib.skgs_binning_buffer_bytes(C.c_int64(max(24 * self.view.P, 64 * T))),), dtype=torch.uint8,
                                      device=st.binning.device)
        keep = (st.tile_bucket, st.binning, st._bufs)
        st.tile_bucket, st.binning = 0, self._spare
        st._bufs = _C._buffers(st.geom, st.binning, st.img)
        try:
            st.forward(None, None)
            s = _C.read_status(st.geom)
        finally:
            st.tile_bucket, st.binning, st._bufs = keep
        if s['overflow']:
            self._spare = None
            raise RuntimeError(f"fused reference route: {s['num_rendered']} tile instances do not fit the probing buffer")
        return max(int(s['max_tile_count']), 1), int(s['num_rendered'])

    @staticmethod
    def _bucket_for(longest):
        bucket = ((int(longest * 1.5) + 63) // 64) * 64     # the package's own trainer: 50 % head room (benchlib/sk_stage.py)
        if 512 < bucket and longest * 1.2 <= 512:
            bucket = 512                                    # a list one wave sorts needs no merge launch behind it
        return bucket

    def _retune(self):
        """measure the longest tile list and give every tile 1.5 x that many slots; training moves and resizes the Gaussians, so the
        measurement is repeated (RETUNE_AT, then every RETUNE_EVERY calls): the capacity follows the scene in both directions before
        a list can overflow, and a scene whose lists have shrunk below 512 entries loses its merge-sort launch again"""
        longest, R = self._measure_longest()
        want = self._bucket_for(longest)
        have = self._bucket
        self.longest, self.num_rendered = longest, R
        hint = max(1, int(1.5 * R / max(self.view.P, 1) + 0.999))       # tiles a Gaussian touches, with head room: the scatter launch's lanes
        lanes = lambda h: 4 if h <= 24 else 8 if h <= 64 else 16           # noqa: E731  (csrc/binning.hip::scatter_lanes)
        old = getattr(self.step, 'tiles_per_gaussian_hint', 0)
        if old == 0 or lanes(old) != lanes(hint):
            self.step.tiles_per_gaussian_hint, self.graphs = hint, None
        if want == have or (want < have and want > 512 and want > 0.75 * have):
            return
        st = self.step
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        st.tile_bucket = want
        st.binning = torch.empty((self.lib.skgs_binning_buffer_bytes(C.c_int64(T * want)),), dtype=torch.uint8, device=st.binning.device)
        from sk_gs_amd import _C
        st._bufs = _C._buffers(st.geom, st.binning, st.img)
        st.geom[:256].zero_()
        self._bucket, self._seen_events, self.graphs = want, 0, None
        calls['capacity_retuned'] += 1

    def _graph_state(self):
        t = self.table.settings
        return (int(t.sh_degree), float(t.scale_modifier), self._bucket, self.step.binning.data_ptr())

    def _capture(self):
        """the forward half and the backward half as ONE hipGraph each (every pointer they bake in is persistent: parameters, the
        persistent gradients, the live slot, the background, the target and cotangent buffers); what still runs per call on the host:
        the slot fill, two replays, the loss launches.  Re-captured when the SH degree, the scale modifier or the binning buffer change;
        a capture that fails leaves the eager launches in place."""
        from sk_gs_amd.train_step import GraphedSteps
        st = self.step
        saved = [(p, p.grad) for p, _, _ in self._attach]
        for p, g, _ in self._attach:
            p.grad = g
        try:
            def fwd(_):
                self._fwd = st.forward(None, None)

            def bwd(k):
                self._launch_backward(extras=(k == 'bx'))
            st.dL_dimage.zero_()      # (the warm-up execution of the backward half runs on it)
            gs = GraphedSteps(lambda k: (fwd if k == 'f' else bwd)(k), warmup=1, collect_garbage=False, thread_local=True)
            gs.capture('f')
            gs.capture('b')
            if self.stage == 'sp':    # the same with the cotangents on _knn_w / _spT read from their persistent buffers
                gs.capture('bx')
            self.graphs, self._graph_key = gs, self._graph_state()
        except Exception as e:   # noqa: BLE001  (no graph: the same launches, issued one by one)
            warnings.warn(f'fused reference route: hipGraph capture failed ({type(e).__name__}: {e}); using eager launches')
            self.graphs, self.use_graphs = None, False
        finally:
            for p, g in saved:
                p.grad = g

    def _launch_backward(self, extras=False):
        st = self.step
        with torch.no_grad():
            if self.stage == 'sp':
                st.g_weights_extra, st.g_bone_T_extra = (self.gw_extra, self.gT_extra) if extras else (None, None)
            st._zero_table_grads()
            st._raster_backward(self._fwd[0], self._fwd[1], None)
            st.backward_skinning(None)

    def check_status(self):
        st = self.step.status()
        if st.get('mlp_failed', 0):
            raise RuntimeError(f"fused reference route: {st['mlp_failed']} skeleton launches gave up their in-launch exchange")
        if st.get('pairs_overflow_events', 0):
            raise RuntimeError(f"fused reference route: a superpoint's inverse neighbour list overflowed in {st['pairs_overflow_events']} steps")
        if st['overflow_events'] > self._seen_events:
            warnings.warn(f"fused reference route: {st['overflow_events'] - self._seen_events} forward(s) of the last {CHECK_EVERY} had tile "
                          f"lists longer than the {self._bucket} slots per tile (excess splats were dropped there); capacity doubled")
            grads = {p: p.grad for p in self.view.parameters()}
            for p in self.view.parameters():
                p.grad = self.grads[p]
            self.step.grow_capacity(2.0)
            self._bucket = self.step.tile_bucket
            self.graphs = None
            for p, g in grads.items():
                p.grad = g
            self._seen_events = 0
            calls['capacity_grown'] += 1

    # -------------------------------------------------------------------------------------------------------------------------
    def fill_slot(self, info, t, time_id):
        from sk_gs_amd import _C
        f = lambda x, n: x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()  # noqa: E731
        tw, tc, cp, fov = f(info['Tw2v'], 16), f(info['Tv2c'], 16), f(info['campos'], 3), f(info['FoV'], 2)
        tt = f(t, 1)
        dev_frame, host_frame = None, 0
        if torch.is_tensor(time_id):
            if time_id.is_cuda:
                dev_frame = time_id if time_id.dtype == torch.int64 else time_id.long()
            else:
                host_frame = int(time_id)
        else:
            host_frame = int(time_id)
        _C._check(self.lib.skgs_view_slot_fill(_p(tw), _p(tc), _p(cp), _p(fov), _p(tt), _p(dev_frame), C.c_int32(host_frame), C.c_int32(0),
                                               _p(self.table.slot), _C._stream()))
        self._keep = (tw, tc, cp, fov, tt, dev_frame)      # (alive until the launch has read them)

    def set_background(self, background):
        bg = self.step.background
        if background is None:
            key = None
        else:
            key = (background.data_ptr(), background._version)
        if key == self._bg_key:
            return
        with torch.no_grad():
            if background is None:
                bg.zero_()                                                   # (prepare_inputs: Tw2v.new_zeros(3), gaussian_splatting.py:262)
            else:
                bg.copy_(background.reshape(-1).expand(3))                   # (:265)
        self._bg_key = key

    def render(self, ref, info, t, time_id, background, stage):
        from sk_gs_amd import _C
        self.serial += 1
        self._terms = None
        if self.serial % CHECK_EVERY == 0:
            self.check_status()
        self.fill_slot(info, t, time_id)
        self.set_background(background)
        if self.serial in self.RETUNE_AT or self.serial % self.RETUNE_EVERY == 0:
            self._retune()
        if self.use_graphs and (self.graphs is None or self._graph_key != self._graph_state()):
            self._capture()
        st = self.step
        out = FusedOutputs(self)
        if self.stage == 'sk':
            image = _FusedRender.apply(self, self.view._xyz)                 # [3,H,W]
            out['_knn_w'] = st.weights.unsqueeze(0)
            out['_skT'], out['_sk_rot'], out['_sk_scale'] = st.bone_T.unsqueeze(0), st._d_rot.unsqueeze(0), st._d_scale.unsqueeze(0)
        else:
            # outputs['_knn_w'] and outputs['_spT'] carry gradient in the reference (the `sparse` / `smooth` regularisers and the joint
            # losses read them, sk_gs.py:1555-1574): outputs of the node; their cotangents enter the backward half
            image, knn_w, spT = _FusedRenderSp.apply(self, self.view._xyz)
            out['_knn_w'], out['_spT'] = knn_w.unsqueeze(0), spT.unsqueeze(0)
            out['_sp_scale'] = st.net.d_scale.unsqueeze(0)
            if self.view.sep_rot:                                            # (None without sep_rot: dropped by render's stack, :1236)
                out['_sp_rot'] = st.net.d_rot.unsqueeze(0)
            # calc_LBS_weight's side effect while the skeleton is not initialised (sk_gs.py:771-773): the stage's latest weights
            ref.sp_weights, ref.sp_knn = st.weights, st.indices
            if self.view.warp_method == 'largest':   # sp_stage's own side effect while training (:849-850): the superpoint each Gaussian follows
                with torch.no_grad():
                    ref.p2sp = torch.gather(st.indices, -1, st.weights.argmax(dim=-1, keepdim=True))[:, 0]
        out['images'] = image.permute(1, 2, 0).unsqueeze(0)                  # [1,H,W,3]: torch.permute(images, (1, 2, 0)) stacked (:1229,1240)
        out['viewspace_points'] = [self.vp]
        out['radii'] = st.radii.unsqueeze(0)
        out['points'] = st.means.unsqueeze(0)
        out['_knn_i'] = st.indices.unsqueeze(0)
        out['stage'] = stage
        calls['render_fused'] += 1
        return out

    # -------------------------------------------------------------------------------------------------------------------------
    def attach_grads(self):
        """the parameters' ``.grad`` = the persistent tensors the kernels write (the reference sets them to None after every step,
        my_ext/framework.py:305); returns what has to be added afterwards: gradients another term accumulated before this node ran
        (or ours from an earlier view that nobody cleared)"""
        foreign = []
        for p, g, store in self._attach:
            cur = p.grad
            if cur is None or store:
                p.grad = g
            elif cur is g or cur.data_ptr() == g.data_ptr():
                foreign.append((g, cur.clone()))
            else:
                foreign.append((g, cur))
                p.grad = g
        return foreign

    def backward(self, g_image, g_w=None, g_T=None):
        from sk_gs_amd import _C
        st = self.step
        a, d = self._fwd
        extras = g_w is not None or g_T is not None
        if extras:
            with torch.no_grad():
                self.gw_extra.copy_(g_w.reshape(self.gw_extra.shape)) if g_w is not None else self.gw_extra.zero_()
                self.gT_extra.copy_(g_T.reshape(self.gT_extra.shape)) if g_T is not None else self.gT_extra.zero_()
            calls['backward_extras'] += 1
        if g_image is None:          # (only the regularisers were differentiated: the image's cotangent is zero)
            with torch.no_grad():
                st.dL_dimage.zero_()
            g_image, self._dimage_ready = st.dL_dimage, True
        if not (self._dimage_ready and g_image.data_ptr() == st.dL_dimage.data_ptr() and g_image.is_contiguous()):
            with torch.no_grad():
                st.dL_dimage.copy_(g_image)
            calls['backward_cotangent'] += 1
        else:
            calls['backward_direct'] += 1
        self._dimage_ready = False
        foreign = self.attach_grads()
        with torch.no_grad():
            if self.graphs is not None:
                self.graphs.graphs['bx' if extras else 'b'].replay()
            else:
                self._launch_backward(extras)
            self.vp.grad = st.grad_means2D
            if foreign:
                for ours, theirs in foreign:
                    ours.add_(theirs)
                calls['foreign_grads_added'] += len(foreign)
            if self._tiles is not None:   # only this step's neighbours hold a gradient on the logit table, unless somebody else added one
                spw = self.grads[self.view.sp_W]
                self._tiles.scan = any(ours is spw for ours, _ in foreign)
        for p in self._stores:      # (the heads own these rows as far as any optimizer is concerned)
            p.grad = None

    # -------------------------------------------------------------------------------------------------------------------------
    def image_terms(self, pred, gt):
        """(L1 mean, 1 - SSIM mean) of the rendered image against ``gt`` [...,H,W,3|4]: one launch, one autograd node, cached per call"""
        key = (self.serial, gt.data_ptr(), gt._version, tuple(gt.shape))
        if self._terms is not None and self._terms[0] == key:
            return self._terms[1]
        g = gt
        while g.dim() > 3 and g.shape[0] == 1:
            g = g[0]
        with torch.no_grad():
            self.target.copy_(g[..., :3].permute(2, 0, 1))                 # [3,H,W] (the reference's targets are HWC): one launch
        out = _FusedImageTerms.apply(pred, self)
        self._terms = (key, out)
        calls['image_terms_fused'] += 1
        return out


class FusedOutputs(dict):
    """``render``'s dict; the per-Gaussian blends the fused kernels never materialise are computed on first access"""

    def __init__(self, route):
        super().__init__()
        self._route = route

    def __missing__(self, key):
        st = self._route.step
        with torch.no_grad():
            if key == 'visibility_filter':
                v = (st.radii > 0).unsqueeze(0)
            elif key in ('_d_xyz', '_d_rot', '_d_scale') and self._route.stage == 'sk':
                w, i = st.weights, st.indices
                if key == '_d_xyz':
                    v = st.means - self._route.view._xyz.detach()
                elif key == '_d_rot':
                    v = (st._d_rot[i] * w[..., None]).sum(dim=1)
                else:
                    v = (st._d_scale[i] * w[..., None]).sum(dim=1)
            else:
                raise KeyError(key)
        self[key] = v
        return v


class _FusedRender(torch.autograd.Function):
    """forward half / backward half of the view in the live slot.  The node hands autograd no gradient tensors: the kernels write the
    parameters' ``.grad`` (returning them would make ``AccumulateGrad`` copy ~30 tensors per step)."""

    @staticmethod
    def forward(ctx, route, anchor):
        # (`anchor`: ONE parameter, so that the image requires grad; the node writes every parameter's gradient itself)
        ctx._skgs_route, ctx.serial = route, route.serial
        ctx.set_materialize_grads(False)      # (an output nobody differentiated arrives as None, not as a tensor of zeros)
        if route.graphs is not None:
            route.graphs.graphs['f'].replay()
        else:
            route._fwd = route.step.forward(None, None)
        route._dimage_ready = False
        return route.step.image.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_image):
        route = ctx._skgs_route
        if ctx.serial != route.serial:
            raise RuntimeError('fused reference route: backward of a render() whose buffers a later render() has overwritten '
                               '(one view at a time: call backward before the next render)')
        if g_image is not None:
            route.backward(g_image)
        return None, None


class _FusedRenderSp(torch.autograd.Function):
    """stage sp: the image, the LBS weights [P,K] and the superpoint transforms [M,7] are the node's differentiable outputs"""

    @staticmethod
    def forward(ctx, route, anchor):
        ctx._skgs_route, ctx.serial = route, route.serial
        ctx.set_materialize_grads(False)      # (an output nobody differentiated arrives as None, not as a tensor of zeros)
        if route.graphs is not None:
            route.graphs.graphs['f'].replay()
        else:
            route._fwd = route.step.forward(None, None)
        route._dimage_ready = False
        st = route.step
        return st.image.detach(), st.weights.detach(), st.net.bone_T.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_image, g_w, g_T):
        route = ctx._skgs_route
        if ctx.serial != route.serial:
            raise RuntimeError('fused reference route: backward of a render() whose buffers a later render() has overwritten '
                               '(one view at a time: call backward before the next render)')
        if g_image is not None or g_w is not None or g_T is not None:
            route.backward(g_image, g_w, g_T)
        return None, None


class _FusedImageTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, route):
        from sk_gs_amd import _C
        st, target = route.step, route.target
        ctx.route, ctx.shape, ctx.serial = route, tuple(pred.shape), route.serial
        ctx.set_materialize_grads(False)
        row = route.loss_ring[route._ring_pos]
        route._ring_pos = (route._ring_pos + 1) % route.loss_ring.shape[0]
        # lambdas (0, 1): loss3 = {0 * L1 + 1 * (1 - SSIM), L1 mean, SSIM mean}
        _C._check(route.lib.skgs_image_loss_forward(C.c_int32(3), C.c_int32(st.H), C.c_int32(st.W), _p(st.image), _p(target), None,
                                                    C.c_float(0.0), C.c_float(1.0), _p(row), _p(st.loss_ws), C.c_size_t(st.loss_ws.numel()),
                                                    _C._stream()))
        return row[1].detach(), row[0].detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_l1, g_ssim):
        from sk_gs_amd import _C
        route, st = ctx.route, ctx.route.step
        if ctx.serial != route.serial:
            raise RuntimeError('fused reference route: backward of image terms whose render() has been overwritten by a later one')
        if g_l1 is None and g_ssim is None:
            return None, None
        fix = lambda g: None if g is None else g.detach().reshape(1).to(torch.float32).contiguous()  # noqa: E731
        gl, gs = fix(g_l1), fix(g_ssim)
        _C._check(route.lib.skgs_image_loss_backward_terms(C.c_int32(3), C.c_int32(st.H), C.c_int32(st.W), _p(st.image), _p(route.target), None,
                                                           _p(gl), _p(gs), _p(st.loss_ws), C.c_size_t(st.loss_ws.numel()), _p(st.dL_dimage),
                                                           _C._stream()))
        route._dimage_ready = True
        g = st.dL_dimage                                                    # [3,H,W] -> the shape `pred` came in
        shape = ctx.shape
        if shape[-1] == 3 and shape[-3:] != (3, st.H, st.W):
            g = g.permute(1, 2, 0)
        return g.reshape(shape), None


# ------------------------------------------------------------------------------------------------ recognising a fused image
def route_of_model(model, stage='sk'):
    """the route built for ``model`` in ``stage`` (None: none yet, or the conditions refused it -- ``why_not['render']``)"""
    r = (_routes.get(model) or {}).get(stage)
    return None if (r is None or isinstance(r, tuple)) else r


def route_of(image):
    """the route whose render node produced ``image`` (through view operations only), or None"""
    fn, hops = getattr(image, 'grad_fn', None), 0
    while fn is not None and hops < 8:
        r = getattr(fn, '_skgs_route', None)
        if r is not None:
            st = r.step
            if fn.serial == r.serial and image.data_ptr() == st.image.data_ptr() and image.numel() == st.image.numel():
                return r
            return None
        if not type(fn).__name__.startswith(_VIEW_NODES) or len(fn.next_functions) != 1:
            return None
        fn, hops = fn.next_functions[0][0], hops + 1
    return None


def _hwc_of(route, pred):
    """``pred`` is the route's whole image, laid out [..1,] H, W, 3 (as ``loss`` hands it over) or [..1,] 3, H, W"""
    st = route.step
    s = tuple(pred.shape)
    while len(s) > 3 and s[0] == 1:
        s = s[1:]
    return s == (st.H, st.W, 3) or s == (3, st.H, st.W)


def image_loss_forward(self, pred_image, gt_image, mask=None):
    """``ImageLoss.forward`` (networks/losses/image_loss.py:19-32): the L1 term of a fused render comes out of the fused loss launch"""
    from sk_gs_amd import reference_accel as ra
    if getattr(self, 'method', None) == 'l1' and not getattr(self, 'masked', False) and mask is None and torch.is_tensor(pred_image):
        route = route_of(pred_image)
        if route is not None and _hwc_of(route, pred_image) and torch.is_tensor(gt_image) and gt_image.is_cuda \
                and gt_image.shape[-3:-1] == pred_image.shape[-3:-1] and gt_image.shape[-1] in (3, 4) and pred_image.shape[-1] == 3:
            return route.image_terms(pred_image, gt_image)[0]
    calls['image_terms_reference'] += 1
    return ra._originals['image_loss'](self, pred_image, gt_image, mask)


def ssim_terms(self, img1, img2):
    """the SSIM term of a fused render, or None (``reference_accel.ssim_loss_forward`` then takes its own paths)"""
    if getattr(self, 'window_size', 11) == 11 and getattr(self, 'reduction', 'mean') == 'mean' and torch.is_tensor(img1):
        route = route_of(img1)
        if route is not None and _hwc_of(route, img1) and torch.is_tensor(img2) and img2.is_cuda and img1.shape[-1] == 3 \
                and img2.shape[-3:-1] == img1.shape[-3:-1] and img2.shape[-1] in (3, 4):
            return route.image_terms(img1, img2)[1]
    return None


# ------------------------------------------------------------------------------------------------ SkeletonGaussianSplatting.render
def _size_of(info, cached):
    """(W, H) of ``info['size']``: ints as the loaders hand them; entries that are DEVICE tensors (``tensor_to`` moved them) would cost
    a read-back per call -- the reference's own settings tuple pays it -- so a route that exists answers with the size it was built for
    (one image size per data set)"""
    s = info['size']
    w, h = s[0], s[1]
    if cached is not None and not isinstance(cached, tuple) and (torch.is_tensor(w) and w.is_cuda):
        return cached.W, cached.H
    return int(w), int(h)


def _route_for(self, stage, t, info, background, time_id, scale_modifier, args, kwargs):
    if stage not in ('sk', 'sp'):
        return None, f'stage {stage!r} (the fused route covers sk and sp)'
    if not (self.training and torch.is_grad_enabled()):
        return None, 'not training / grad disabled'
    if 'hook' in kwargs:      # (the only keyword the reference's render reads besides its named ones, sk_gs.py:1222-1223: it edits the
        return None, 'a hook on the network outputs'      # per-Gaussian tensors the fused kernels never materialise; other keywords --
                                                           # rays_o / rays_d of a loader with rays -- are ignored there and here)
    if t is None or time_id is None or not torch.is_tensor(t) or not t.is_cuda or t.numel() != 1:
        return None, 't / time_id: one frame of the training set on the device'
    if torch.is_tensor(time_id) and time_id.numel() != 1:
        return None, 'time_id'
    need = ('Tw2v', 'Tv2c', 'campos', 'FoV', 'size')
    if not all(k in info for k in need) or not all(torch.is_tensor(info[k]) and info[k].is_cuda for k in need[:4]):
        return None, 'info tensors are not on the device'
    if info['Tw2v'].numel() != 16 or info['Tv2c'].numel() != 16 or info['campos'].numel() != 3 or info['FoV'].numel() != 2:
        return None, 'more than one view per call'
    if background is not None and not (torch.is_tensor(background) and background.is_cuda and background.numel() <= 3):
        return None, 'an image-shaped background'
    per_stage = _routes.get(self)
    if per_stage is None:
        per_stage = _routes[self] = {}
    cached = per_stage.get(stage)
    W, H = _size_of(info, cached)
    sh = int(self.active_sh_degree) if not hasattr(self, '_active_sh_degree') else _cached_sh_degree(self)
    if isinstance(cached, tuple):          # (reason, light identity): no route for this model as it is
        if cached[1] == _light_identity(self, stage):
            return None, cached[0]
        cached = None
    if cached is not None and (cached.light != _light_identity(self, stage) or (cached.W, cached.H) != (W, H)):
        cached = None
    if cached is None:
        reason = _conditions(self) if stage == 'sk' else _conditions_sp(self)
        if reason is not None:
            per_stage[stage] = (reason, _light_identity(self, stage))
            return None, reason
        cached = per_stage[stage] = FusedReferenceRoute(self, W, H, sh, scale_modifier, stage)
        cached.light = _light_identity(self, stage)
    cached.table.settings.sh_degree, cached.table.settings.scale_modifier = sh, float(scale_modifier)
    return cached, None


_LIGHT = {'sk': ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_W', 'joints', 'global_tr', 'sk_cache',
                 'joint_parents', 'sk_deform_net', 'sk_is_init'),
          'sp': ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_W', 'sp_points', 'hyper_feature',
                 'sp_hyper_feature', '_sp_radius', '_sp_weight', 'sp_deform_net')}


def _light_identity(ref, stage='sk'):
    """what can change under a route between two calls, cheap enough for every call: the objects behind the attributes the step reads
    (densification / re-initialisation REPLACE the Parameters, set_from_dataset the tables), the storage of the first, the topology's
    version, the skeleton flag's"""
    obj = tuple(id(getattr(ref, n, None)) for n in _LIGHT[stage])
    x = ref._xyz
    if stage == 'sp':
        return obj + (x.data_ptr(), int(x.shape[0]), int(ref.sp_points.shape[0]), int(getattr(ref, 'num_knn', 0)),
                      getattr(ref, 'LBS_method', None), getattr(ref, 'warp_method', None), bool(getattr(ref, 'sep_rot', False)))
    jp, flag = getattr(ref, 'joint_parents', None), getattr(ref, 'sk_is_init', None)
    return obj + (x.data_ptr(), int(x.shape[0]), None if jp is None else jp._version, None if flag is None else flag._version,
                  int(getattr(ref, 'num_knn', 0)), getattr(ref, 'LBS_method', None))


_sh_cache = weakref.WeakKeyDictionary()


def _cached_sh_degree(ref):
    """``active_sh_degree`` is ``self._active_sh_degree.item()`` (gaussian_splatting.py:185-190: a read-back); the setter REPLACES the
    buffer, so the value is cached per buffer object"""
    buf = ref._active_sh_degree
    hit = _sh_cache.get(ref)
    if hit is None or hit[0] is not buf or hit[1] != buf._version:
        hit = _sh_cache[ref] = (buf, buf._version, int(buf.item()))
    return hit[2]


def render(self, *args, t=None, info, background=None, time_id=None, scale_modifier=1.0, stage=None, **kwargs):
    """``SkeletonGaussianSplatting.render`` (networks/sk_gs.py:1206-1242) with the fused step behind it when its conditions hold"""
    from sk_gs_amd import reference_accel as ra
    stage = self.get_now_stage(stage)
    route, reason = _route_for(self, stage, t, info, background, time_id, scale_modifier, args, kwargs)
    if route is None:
        why_not['render'] = reason
        calls['render_reference'] += 1
        return ra._originals['render'](self, *args, t=t, info=info, background=background, time_id=time_id, scale_modifier=scale_modifier,
                                       stage=stage, **kwargs)
    return route.render(self, info, t, time_id, background, stage)
