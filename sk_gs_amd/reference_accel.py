"""Optional accelerators for an UNMODIFIED checkout of the reference, applied AFTER it has been imported::

    import sk_gs_amd
    sk_gs_amd.install_reference_hooks()          # before `import networks`: the compiled ops and the two third-party stand-ins
    import networks, train
    sk_gs_amd.accelerate_reference()             # after: five methods of the reference's classes (+ torch.optim.Adam.step) get a fast path

``install_reference_hooks`` makes the reference RUN on an MI355X; with it alone five pieces of its training step are still long chains
of small torch launches.  ``accelerate_reference`` replaces exactly those methods -- same arguments, same returned objects, and a
call that does not match the fast path's conditions is handed to the reference's own method:

* ``networks.losses.ssim.SSIM_Loss.forward`` (ssim.py:26-43: five depth-wise 11x11 convolutions + ~30 element-wise kernels, 3.5 ms per
  800x800 image on this GPU, forward + backward) -> the fused kernels of ``sk_gs_amd.losses.image_loss`` (``skgs_image_loss_forward /
  _backward``: 41 us) for one fp32 3-channel image pair on a HIP device, window 11, reduction 'mean';
* ``networks.sk_gs.SkeletonGaussianSplatting.kinematic`` (sk_gs.py:1069-1107: ``sk_deform_net`` -> ``SO3`` -> ``sk_t = joints +
  sk_r.act(-joints)`` -> ``skeleton_warp_SE3``, ~60 Lie-group launches forward and backward through the lietorch stand-in) -> the
  reference's own network call followed by ONE launch per direction, ``skgs_bone_chain_forward / _backward``
  (``sk_gs_amd.skeleton.bone_chain``), for the training-time quaternion path (``which_rotation: quaternion``, no ``sk_r_delta``, no
  ``sk_feature``, a 7-vector or no global transform).  The result is handed back as ``SE3.InitFromVec(T)`` of the stand-in, so
  everything the reference does with ``sk_T`` afterwards -- ``sk_T[indices].act(...)``, ``sk_T.vec()`` -- is unchanged; the train-time
  cache write ``self.sk_cache[time_id] = ...`` (:1077-1079) is kept.

* ``networks.sk_gs.SimpleDeformationNetwork.forward`` (sk_gs.py:158-164: two frequency encoders + ``MLP_with_skips``, ~65 torch launches
  forward and ~130 backward for the 20 joint rows of stage `sk`) -> the persistent one-launch-per-direction network kernels
  (``skgs_deform_mlp_forward / _backward``, ``sk_gs_amd.deform_net``) on a SHADOW module whose hidden-layer parameters ARE the
  reference module's ``nn.Parameter`` objects (the heads are concatenated into one persistent matrix per call; autograd routes their
  gradient back through that concatenation) -- for <= 48 rows, one time for all rows, width 256;
* ``networks.sk_gs.DeformNetwork.forward`` (sk_gs.py:295-317: the superpoint stage's network on the 512 superpoints) -> the MFMA
  row-block kernels (``skgs_sp_net_forward / _backward``, ``sk_gs_amd.superpoint.SpDeformNet``) on a shadow that shares EVERY
  parameter object (the two classes have the same parameter names), ``is_blender`` either way, with or without the ``local_rotation``
  head; the returned dict has ``d_xyz / d_rotation / d_scaling (/ g_rotation)`` -- not ``hidden``, which nothing in the reference reads.

* ``networks.sk_gs.SkeletonGaussianSplatting.calc_LBS_weight`` (sk_gs.py:751-774) -> ``sk_gs_amd.deform.calc_lbs_weight``: the search
  and the weighting (kernel / weighted kernel / `W` logits / distance softmax, 3 or 3 + 8 search dimensions) as one launch per direction.

* ``networks.renderer.gaussian_render_origin.render_gs_offical`` (the adapter to the upstream rasterizer API the shipped configs render
  through; also under the name ``networks.gaussian_splatting`` bound at import -- call ``accelerate_reference()`` BEFORE the model is
  built, ``GaussianSplatting.__init__`` stores the function) -> the same function, called with typed rotations whose swizzle
  ``rotations[..., (3, 0, 1, 2)]`` is two slices instead of an index list (0.30 ms of backward per step).
* ``torch.optim.Adam.step`` (the optimizer the reference builds, gaussian_splatting.py:443-460: ten parameter groups, ``eps=1e-15``) ->
  ONE launch of ``skgs_adam_step_range`` over a descriptor table of the optimizer's OWN state tensors (the reference's per-parameter
  state surgery keeps working); torch's foreach form is ~80 launches per step for those groups.  The one patch outside the reference's
  classes: ``accelerate_reference(adam=False)`` leaves torch alone.

The reference's files are not touched; ``restore_reference()`` puts the original methods back.
"""
from __future__ import annotations

import sys
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

_originals = {}
calls = {'ssim_fused': 0, 'ssim_reference': 0, 'kinematic_fused': 0, 'kinematic_reference': 0, 'sk_net_fused': 0, 'sk_net_reference': 0,
         'sp_net_fused': 0, 'sp_net_reference': 0, 'lbs_weight_fused': 0, 'lbs_weight_reference': 0, 'adam_fused': 0, 'adam_reference': 0, 'swizzle_fused': 0,
         'weight_reg_fused': 0, 'weight_reg_reference': 0, 'adam_tiled': 0}  # counters (tests)


# ------------------------------------------------------------------------------------------------ SSIM_Loss.forward
def _as_chw(img: torch.Tensor):
    """one image as [3,H,W] contiguous, or None when it is not one fp32 3-channel image on a HIP device"""
    if not (img.is_cuda and img.dtype == torch.float32):
        return None
    if img.dim() == 4 and img.shape[0] == 1:
        img = img[0]
    if img.dim() != 3:
        return None
    if img.shape[-1] == 3 and img.shape[0] != 3:      # HWC, as sk_gs.loss hands it over (sk_gs.py:1527-1529)
        return img.permute(2, 0, 1).contiguous()
    if img.shape[0] == 3:
        return img.contiguous()
    return None


def ssim_loss_forward(self, img1, img2):
    """``SSIM_Loss.forward`` (networks/losses/ssim.py:26-43): ``1 - mean SSIM`` of one image pair through the fused kernels"""
    if 'render' in _originals:      # an image out of the fused render node: the term comes from the one fused loss launch
        from sk_gs_amd import reference_fused as rf
        term = rf.ssim_terms(self, img1, img2)
        if term is not None:
            calls['ssim_fused'] += 1
            return term
    if getattr(self, 'window_size', 11) == 11 and getattr(self, 'reduction', 'mean') == 'mean' and img1.shape == img2.shape:
        a, b = _as_chw(img1), _as_chw(img2)
        if a is not None and b is not None:
            from sk_gs_amd.losses import image_loss
            calls['ssim_fused'] += 1
            return image_loss(a, b, 0.0, 1.0)     # lambda_l1 * L1 + lambda_ssim * (1 - SSIM) with (0, 1)
    calls['ssim_reference'] += 1
    return _originals['ssim'](self, img1, img2)


# ------------------------------------------------------------------------------------------------ SkeletonGaussianSplatting.kinematic
_topo_cache = {}    # id(joint_parents tensor) -> (weak reference to it, its version, the root object, its version, topology)


def _topology(parents_table: torch.Tensor, root) -> dict:
    """``build_topology`` of the skeleton in ``joint_parents`` (column 0: the direct parent, sp_gs_joint.cu:55-85) / ``joint_root``,
    cached per table OBJECT until the table is rewritten in place or replaced (joint discovery runs every 1000+ iterations).  (ADVICE r5:
    a key of address + version could alias across model instances -- another model's table allocated where a freed one lay.)"""
    from sk_gs_amd.skeleton import build_topology
    hit = _topo_cache.get(id(parents_table))
    rv = root._version if torch.is_tensor(root) else root
    if hit is None or hit[0]() is not parents_table or hit[1] != parents_table._version or hit[2] is not root or hit[3] != rv:
        for k in [k for k, v in _topo_cache.items() if v[0]() is None]:     # (tables that are gone)
            del _topo_cache[k]
        r = int(root.reshape(-1)[0]) if torch.is_tensor(root) else int(root)
        hit = _topo_cache[id(parents_table)] = (weakref.ref(parents_table), parents_table._version, root, rv,
                                                 build_topology(parents_table[:, 0].long().cpu(), r, parents_table.device))
    return hit[4]


_bias_cache = {}


def _quat_bias(like: torch.Tensor) -> torch.Tensor:
    """[0, 0, 0, 1] on ``like``'s device, built once (``x.new_tensor([...])`` is a blocking host-to-device copy per call)"""
    key = (like.device, like.dtype)
    b = _bias_cache.get(key)
    if b is None:
        b = _bias_cache[key] = torch.tensor([0., 0., 0., 1.], dtype=like.dtype, device=like.device)
    return b


def kinematic(self, joints, t, g_tr=None, time_id=None, sk_r_delta=None):
    """``SkeletonGaussianSplatting.kinematic`` (networks/sk_gs.py:1069-1107) with the Lie-group chain as one launch per direction"""
    lie = sys.modules.get('lietorch')
    fast = (joints.is_cuda and joints.dtype == torch.float32 and sk_r_delta is None and getattr(self, 'sk_feature', None) is None
            and (self.training or not self.test_time_interpolate) and getattr(self, '_R_dim', 4) == 4 and lie is not None
            and hasattr(lie, 'fused_calls')                                   # this package's stand-in, not upstream lietorch
            and (g_tr is None or (torch.is_tensor(g_tr) and g_tr.dim() == 1 and g_tr.shape[0] == 7))
            and self.joint_parents.dim() == 2 and joints.shape[0] <= 512)
    if not fast:
        calls['kinematic_reference'] += 1
        return _originals['kinematic'](self, joints, t, g_tr, time_id, sk_r_delta)
    from sk_gs_amd.skeleton import bone_chain
    sk_r_raw, d_rot, d_scale = self.sk_deform_net(joints, t)                  # the reference's own network call (:1074)
    if sk_r_raw.shape[-1] != 4:
        calls['kinematic_reference'] += 1
        return _originals['kinematic'](self, joints, t, g_tr, time_id, sk_r_delta)
    if self.training and time_id is not None:                                  # the train-time cache (:1077-1079)
        with torch.no_grad():
            sk_r = F.normalize(sk_r_raw + _quat_bias(sk_r_raw), dim=-1)
            self.sk_cache[time_id] = torch.cat([sk_r, d_rot, d_scale], dim=-1)
    topo = _topology(self.joint_parents, self.joint_root)
    T = bone_chain(sk_r_raw, joints, g_tr, topo)                               # [M,7] = (t, q_xyzw): kinematic + skeleton_warp_SE3
    calls['kinematic_fused'] += 1
    return lie.SE3.InitFromVec(T), d_rot, d_scale


# ------------------------------------------------------------------------------------------------ SimpleDeformationNetwork.forward
_shadows = weakref.WeakKeyDictionary()   # reference module -> shadow module of this package (parameters shared)


def _freq_degree(enc, input_dim):
    """degree of a reference FreqEncoder (networks/encoders/freq_encoder.py:61-75), or None for any other encoder"""
    d = getattr(enc, 'degree', None)
    if not isinstance(d, int) or getattr(enc, 'input_dim', None) != input_dim:
        return None
    return d if getattr(enc, 'output_dim', None) == input_dim * (1 + 2 * d) else None


def sk_net_shadow(ref):
    """``sk_gs_amd.deform_net.DeformMLP`` over the parameters of a reference ``SimpleDeformationNetwork``, or None when the module is
    not the shape the kernels are written for (the caller then uses the reference's own forward)"""
    hit = _shadows.get(ref)
    net = getattr(ref, 'dynamic_net', None)
    if hit is not None and hit.dynamic_net.last_weight.device == net.net[0].weight.device:
        return hit
    from sk_gs_amd.deform_net import DeformMLP
    p_deg, t_deg = _freq_degree(getattr(ref, 'pos_enc_p', None), 3), _freq_degree(getattr(ref, 'pos_enc_t', None), 1)
    last = getattr(net, 'last', None)
    if (p_deg is None or t_deg is None or net is None or not isinstance(last, nn.ModuleList) or getattr(net, 'weight_norm', False)
            or not getattr(net, 'bias', True) or net.dim_hidden != 256 or not (1 <= len(last) <= 4)):
        return None
    out_channels = tuple(int(l.out_features) for l in last)
    sh = DeformMLP(3, 1, out_channels, width=net.dim_hidden, depth=net.num_layers, skips=tuple(net.skips), p_degree=p_deg, t_degree=t_deg)
    if sh.dynamic_net.in_channels != net.in_channels or any(a.weight.shape != b.weight.shape for a, b in zip(sh.dynamic_net.net, net.net)):
        return None
    for mine, theirs in zip(sh.dynamic_net.net, net.net):   # the SAME Parameter objects: updates, .grad and device moves are shared
        mine.weight, mine.bias = theirs.weight, theirs.bias
    dev = net.net[0].weight.device
    sh.dynamic_net.last_weight = nn.Parameter(torch.cat([l.weight.detach() for l in last]).to(dev), requires_grad=False)
    sh.dynamic_net.last_bias = nn.Parameter(torch.cat([l.bias.detach() for l in last]).to(dev), requires_grad=False)
    _shadows[ref] = sh
    return sh


def simple_deform_forward(self, points, t):
    """``SimpleDeformationNetwork.forward`` (networks/sk_gs.py:158-164) as one launch per direction"""
    from sk_gs_amd.deform_net import _DeformMLPFn, fused_supported
    sh = None
    if (torch.is_tensor(points) and points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3
            and torch.is_tensor(t) and t.numel() == 1):
        sh = sk_net_shadow(self)
    if sh is None or not fused_supported(sh, points.shape[0]):
        calls['sk_net_reference'] += 1
        return _originals['sk_net'](self, points, t)
    net, heads = sh.dynamic_net, self.dynamic_net.last
    if getattr(sh, '_heads_rehomed', None) is None:   # (re-homed by the fused route: the heads ARE rows of that matrix already)
        with torch.no_grad():  # the kernels read ONE head matrix: the current values of the reference's heads, in their order
            torch.cat([h.weight for h in heads], out=net.last_weight.data)
            torch.cat([h.bias for h in heads], out=net.last_bias.data)
    # (what autograd differentiates: the same concatenations -- the gradient of the fused head matrix is split back onto the heads)
    params = [p for l in net.net for p in (l.weight, l.bias)] + [torch.cat([h.weight for h in heads]), torch.cat([h.bias for h in heads])]
    out = _DeformMLPFn.apply(sh, torch.is_grad_enabled(), points, t.to(points.device), *params)
    calls['sk_net_fused'] += 1
    return list(out.split(net.out_channels, dim=-1))


# ------------------------------------------------------------------------------------------------ DeformNetwork.forward
def sp_net_shadow(ref):
    """``sk_gs_amd.superpoint.SpDeformNet`` sharing every parameter of a reference ``DeformNetwork``, or None"""
    hit = _shadows.get(ref)
    if hit is not None:
        return hit
    from sk_gs_amd.superpoint import SpDeformNet
    p_deg, t_deg = _freq_degree(getattr(ref, 'pos_enc_p', None), 3), _freq_degree(getattr(ref, 'pos_enc_t', None), 1)
    if (p_deg != 10 or t_deg is None or getattr(ref, 'D', 0) != 8 or getattr(ref, 'W', 0) != 256 or list(getattr(ref, 'skips', [])) != [4]
            or getattr(ref, 'max_d_scale', -1) > 0):
        return None
    blender = bool(getattr(ref, 'is_blender', False))
    if blender and (getattr(ref, 'time_out', 0) != 30 or tuple(ref.timenet[0].weight.shape) != (256, 1 + 2 * t_deg)):
        return None
    sh = SpDeformNet(t_degree=t_deg, sep_rot=bool(getattr(ref, 'sep_rot', False)), is_blender=blender)
    theirs = dict(ref.named_parameters())
    if set(theirs) != set(dict(sh.named_parameters())) or not sh.kernel_supported():
        return None
    for name in list(theirs):   # the SAME Parameter objects under the same names
        mod, leaf = sh, name.split('.')
        for part in leaf[:-1]:
            mod = getattr(mod, part)
        if getattr(mod, leaf[-1]).shape != theirs[name].shape:
            return None
        setattr(mod, leaf[-1], theirs[name])
    _shadows[ref] = sh
    return sh


SP_NET_MAX_ROWS = 4096   # (a runner keeps 8.6 KB of activations per row and is kept per row count: the superpoint-sized calls only)


def deform_network_forward(self, x, t, **kwargs):
    """``DeformNetwork.forward`` (networks/sk_gs.py:295-317) on the MFMA row-block kernels"""
    sh = None
    if (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3 and not kwargs
            and 1 <= x.shape[0] <= SP_NET_MAX_ROWS and torch.is_tensor(t) and t.numel() == 1):
        sh = sp_net_shadow(self)
    if sh is None:
        calls['sp_net_reference'] += 1
        return _originals['sp_net'](self, x, t, **kwargs)
    calls['sp_net_fused'] += 1
    return sh(x, t.to(x.device))


# ------------------------------------------------------------------------------------------------ calc_LBS_weight
def calc_LBS_weight(self, points, sp_points, feature=None, sp_feature=None, K=None, temperature=1.):
    """``SkeletonGaussianSplatting.calc_LBS_weight`` (networks/sk_gs.py:751-774): the search AND the weighting as one launch per
    direction (``sk_gs_amd.deform.calc_lbs_weight``; the reference's lines are a search + ~10 element-wise / gather launches on
    [P, K] tensors, and as many again backward).  Same priority of the weightings (:760-770), same side effect (:771-773); the
    indices come back typed as the ``knn_points`` stand-in types them (the reference's own ``table[indices]`` gathers keep their
    duplicate-free backward)."""
    ok = (torch.is_tensor(points) and points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 3
          and torch.is_tensor(sp_points) and sp_points.is_cuda and sp_points.dim() == 2 and sp_points.shape[1] == 3
          and (feature is None) == (sp_feature is None) and isinstance(temperature, (int, float)))
    K = self.num_knn if K is None else K
    if not ok or K > 16 or K > sp_points.shape[0]:
        calls['lbs_weight_reference'] += 1
        return _originals['lbs_weight'](self, points, sp_points, feature, sp_feature, K, temperature)
    from sk_gs_amd import pytorch3d_ops as p3d
    from sk_gs_amd.deform import calc_lbs_weight
    kernel = self._sp_radius is not None
    weights, indices = calc_lbs_weight(
        points, sp_points, int(K), sp_W=None if kernel else self.sp_W, kernel_radius=self.kernel_radius if kernel else None,
        kernel_weight=self.kernel_weight if (kernel and self._sp_weight is not None) else None, temperature=float(temperature),
        feature=feature, sp_feature=sp_feature)
    if p3d._TYPED_INDEX:
        indices = p3d.NeighbourIndex.wrap(indices)
    if not self.sk_is_init:
        self.sp_weights = weights.detach()
        self.sp_knn = indices.detach()
    calls['lbs_weight_fused'] += 1
    return weights, indices


# ------------------------------------------------------------------------------------------------ the weight regularisers of stage sp
def loss_weight_sparsity(self, weight, eps=1e-7):
    """``SkeletonGaussianSplatting.loss_weight_sparsity`` (networks/sk_gs.py:1339-1340) as one launch (``skgs_weight_sparsity``)"""
    from sk_gs_amd import weight_reg as wr
    if wr.sparsity_supported(weight) and isinstance(eps, float):
        calls['weight_reg_fused'] += 1
        return wr.weight_sparsity(weight, eps)
    calls['weight_reg_reference'] += 1
    return _originals['w_sparse'](self, weight, eps)


def loss_weight_smooth(self, weight):
    """``SkeletonGaussianSplatting.loss_weight_smooth`` (networks/sk_gs.py:1357-1359): the reference's own ``update_gs_knn()`` (the
    neighbour table of the Gaussians, rebuilt by its own schedule), then value and gradient as one launch (``skgs_weight_smooth``)
    instead of the [P, 21, K] gather and its sort-based index backward"""
    from sk_gs_amd import weight_reg as wr
    self.update_gs_knn()
    nbr = getattr(self, 'gs_knn_index', None)
    if wr.smooth_supported(weight, nbr):
        calls['weight_reg_fused'] += 1
        return wr.weight_smooth(weight, nbr)
    calls['weight_reg_reference'] += 1
    return _originals['w_smooth'](self, weight)


# ------------------------------------------------------------------------------------------------ render_gs_offical
class QuatXYZW(torch.Tensor):
    """the rotations handed to the reference's upstream-rasterizer adapter, typed so that ITS swizzle ``rotations[..., (3, 0, 1, 2)]``
    (gaussian_render_origin.py:41-42) is formed from two slices: the backward of an index LIST is torch's sort-based ``index_put``
    (0.30 ms for 100k quaternions on this GPU, a fifth of the accelerated iteration's GPU time), the backward of slices two strided
    copies.  Only that one expression is changed -- the adapter's code runs as it is; every other use sees and returns plain tensors."""

    @classmethod
    def wrap(cls, q):
        return q.as_subclass(cls) if (torch.is_tensor(q) and q.dim() >= 1 and q.shape[-1] == 4 and type(q) is torch.Tensor) else q

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], cls):
                idx = args[1]
                if (isinstance(idx, tuple) and len(idx) == 2 and idx[0] is Ellipsis and isinstance(idx[1], (tuple, list))
                        and tuple(idx[1]) == (3, 0, 1, 2)):
                    x = args[0].as_subclass(torch.Tensor)
                    calls['swizzle_fused'] += 1
                    return torch.cat([x[..., 3:], x[..., :3]], dim=-1)
            out = func(*[a.as_subclass(torch.Tensor) if isinstance(a, cls) else a for a in args], **kwargs)
            return out


def render_gs_offical(points, opacity, raster_settings, scales=None, rotations=None, *args, **kwargs):
    """``render_gs_offical`` (networks/renderer/gaussian_render_origin.py:11-68) itself, called with typed rotations (``QuatXYZW``)"""
    return _originals['render_adapter'](points, opacity, raster_settings, scales, QuatXYZW.wrap(rotations), *args, **kwargs)


# ------------------------------------------------------------------------------------------------ torch.optim.Adam.step
class _AdamRunner:
    """the device descriptor table of ONE torch.optim.Adam instance for ``skgs_adam_step_range`` (include/skgs.h): built from the
    optimizer's OWN state tensors -- ``state[p]['exp_avg'] / ['exp_avg_sq']`` stay where torch keeps them, so the reference's optimizer
    surgery (gaussian_splatting.py: replace / cat / prune the state per parameter) keeps working -- re-built when any address or size
    changes, the learning rates refreshed (pinned staging, asynchronous copy) when a group's ``lr`` changes"""

    def __init__(self, dev):
        import ctypes as C
        from sk_gs_amd import _C
        self.C, self._C, self.lib = C, _C, _C.load_library()
        self.lib.skgs_adam_state_bytes.restype = C.c_size_t
        self.lib.skgs_adam_chunk_elems.restype = C.c_int64
        self.chunk = int(self.lib.skgs_adam_chunk_elems())
        self.dev = dev
        self.state = torch.zeros(int(self.lib.skgs_adam_state_bytes()) // 4, dtype=torch.float32, device=dev)
        self.key = self.lrs = self.table = self.pin = self.event = None
        self.count = None
        self.n = self.chunks = 0

    RING = 8   # pinned staging buffers in flight: the host may run this many uploads ahead of the device before it has to wait

    def _copy(self, dst, blob):
        """asynchronous upload through a RING of pinned staging buffers (one buffer + ``event.synchronize()`` blocked the
        host on the previous step's copy every time a learning rate changed -- the reference changes two per iteration)"""
        n = len(blob)
        if self.pin is None or self.pin[0].numel() < n:
            self.pin = [torch.empty(max(n, 4096), dtype=torch.uint8, pin_memory=True) for _ in range(self.RING)]
            self.event, self._slot = [None] * self.RING, 0
        k = self._slot
        self._slot = (k + 1) % self.RING
        if self.event[k] is not None:
            self.event[k].synchronize()      # (the upload issued RING uploads ago: long done unless the device is that far behind)
        self.pin[k][:n].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        dst[:n].copy_(self.pin[k][:n], non_blocking=True)
        ev = self.event[k]
        if ev is None:
            ev = self.event[k] = torch.cuda.Event()
        ev.record()

    def step(self, entries, lrs, beta1, beta2, eps, count):
        """entries: [(param, grad, exp_avg, exp_avg_sq)] in table order, lrs: one per entry, count: steps taken so far (all equal).
        An entry whose parameter carries ``_skgs_logit_tiles`` (the dense [P, M] logit table of ``LBS_method: W``, flagged by the fused
        reference route) is updated by ``skgs_adam_masked_rows`` -- the 32-column tiles that have ever held a gradient -- instead of as a
        part of the dense launch: 1.4 GB of optimizer traffic per iteration at P = 1e5, M = 512 otherwise."""
        import struct
        tiled = [(e, lr) for e, lr in zip(entries, lrs) if _tiles_of(e[0]) is not None]
        if tiled:
            dense = [(e, lr) for e, lr in zip(entries, lrs) if _tiles_of(e[0]) is None]
            entries, lrs = [e for e, _ in dense], [lr for _, lr in dense]
        key = (tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, g, m, v in entries),
               tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr)) for (p, g, m, v), lr in tiled))
        if key != self.key or lrs != self.lrs:
            blob, chunk0 = bytearray(), 0
            for (p, g, m, v), lr in zip(entries, lrs):
                blob += struct.pack('<QQQQqqfi', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), chunk0, float(lr), 0)
                chunk0 += (p.numel() + self.chunk - 1) // self.chunk
            n_dense = len(blob)
            for (p, g, m, v), lr in tiled:      # one descriptor each behind the dense table (its own chunk space: chunk0 = 0)
                blob += struct.pack('<QQQQqqfi', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0, float(lr), 0)
            if self.table is None or self.table.numel() < len(blob):
                self.table = torch.zeros(max(len(blob), 56), dtype=torch.uint8, device=self.dev)
            self._copy(self.table, blob)
            self.key, self.lrs, self.n, self.chunks = key, list(lrs), len(entries), chunk0
            self.tiled = []
            for i, ((p, g, m, v), lr) in enumerate(tiled):
                info = _tiles_of(p)
                if info.mask is None or info.mask.numel() != p.shape[0] or info.mask_of is not m:
                    # the live tiles = wherever a moment is non-zero now (an optimizer that has trained the table before)
                    info.mask, info.mask_of = torch.zeros(p.shape[0], dtype=torch.int32, device=self.dev), m
                    self._C._check(self.lib.skgs_adam_logit_mask_rebuild(self.C.c_int32(p.shape[0]), self.C.c_int32(p.shape[1]), self.C.c_void_p(m.data_ptr()),
                                                                        self.C.c_void_p(v.data_ptr()), self.C.c_void_p(info.mask.data_ptr()), self._C._stream()))
                self.tiled.append((p, info, n_dense + 56 * i))
        if count != self.count:   # first fused step, or torch's own step ran in between: the counter and 1 - beta^count re-derived
            import struct as _s
            blob = _s.pack('<ffdd', float(count), 0.0, 1.0 - beta1 ** count, 1.0 - beta2 ** count)
            self.state.zero_()
            self._copy(self.state.view(torch.uint8), blob)
        self.launch(beta1, beta2, eps)
        self.count = count + 1

    tiled = ()

    def launch(self, beta1, beta2, eps):
        """the tiled pieces (they read the counter and the bias corrections of the step in progress), then the dense launch, which
        advances the counter"""
        C = self.C
        for p, info, offset in self.tiled:
            idx = info.indices if (info.indices is not None and info.indices.shape[0] == p.shape[0]) else None
            scan = info.scan or idx is None
            info.scan = True      # (whoever knows that only `indices` received a gradient says so per step: reference_fused.py)
            self._C._check(self.lib.skgs_adam_masked_rows(
                C.c_int32(p.shape[0]), C.c_int32(p.shape[1]), C.c_int32(0 if idx is None else idx.shape[1]), C.c_void_p(None if idx is None else idx.data_ptr()),
                C.c_int32(1 if scan else 0), C.c_void_p(self.table.data_ptr() + offset), C.c_void_p(info.mask.data_ptr()), C.c_double(beta1),
                C.c_double(beta2), C.c_double(eps), C.c_void_p(self.state.data_ptr()), C.c_int32(0), self._C._stream()))
            calls['adam_tiled'] += 1
        self._C._check(self.lib.skgs_adam_step_range(
            C.c_int32(self.n), C.c_void_p(self.table.data_ptr()), C.c_int64(0), C.c_int64(self.chunks), C.c_double(beta1),
            C.c_double(beta2), C.c_double(eps), C.c_void_p(self.state.data_ptr()), C.c_int32(1), None, C.c_int64(0), self._C._stream()))


class LogitTiles:
    """what the fused reference route hangs on the dense [P, M] logit table (``param._skgs_logit_tiles``): ``indices`` [P, K] = the columns
    this step's gradient can be non-zero in, ``scan`` = something else may have written to the gradient as well (the default: the update
    then looks at the whole gradient row before it trusts the mask), ``mask`` [P] = the live 32-column tiles (built from the moments)"""

    def __init__(self, indices):
        self.indices, self.scan, self.mask, self.mask_of = indices, True, None, None


def _tiles_of(p):
    info = getattr(p, '_skgs_logit_tiles', None)
    if info is None or p.dim() != 2 or p.shape[1] > 1024 or p.shape[0] == 0:
        return None
    return info


_adam_runners = weakref.WeakKeyDictionary()
_adam_plans = weakref.WeakKeyDictionary()


class _AdamPlan:
    """what the last fused step of an optimizer launched, so that the next one -- same parameters in the same groups, the same gradient
    tensors, the same learning rates, every parameter one step older -- is ONE comparison pass and the launch (the per-step walk over
    groups, states and gradients with its ~35 table entries was 0.13 ms of host time per iteration).  Anything that differs (a
    learning rate, a gradient tensor, a parameter list after densification, a step count somebody moved) returns False: the general
    path rebuilds."""

    def __init__(self, opt, runner, entries, count):
        self.runner, self.count = runner, count
        self.params = [p for p, _, _, _ in entries]
        self.grads = [g for _, g, _, _ in entries]
        self.moments = [(m, v) for _, _, m, v in entries]
        self.states = [opt.state[p] for p in self.params]
        self.steps = [st['step'] for st in self.states]
        self.with_grad = set(map(id, self.params))
        self.layout = [(id(g), len(g['params'])) for g in opt.param_groups]
        self.lrs = [g['lr'] for g in opt.param_groups]
        g0 = opt.param_groups[0]
        self.hyper = (float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']))
        self.tiles = tuple(id(_tiles_of(p)) for p, _, _ in runner.tiled)
        self.flagged = [(p, _tiles_of(p)) for p in self.params]

    def replay(self, opt) -> bool:
        groups = opt.param_groups
        if len(groups) != len(self.layout) or self.runner.count != self.count:
            return False
        for g, (gid, n), lr in zip(groups, self.layout, self.lrs):
            if id(g) != gid or len(g['params']) != n or g['lr'] != lr:
                return False
        for p, gr in zip(self.params, self.grads):
            if p.grad is not gr:
                return False
        for p, info in self.flagged:
            if _tiles_of(p) is not info:
                return False
        k = 0
        for g in groups:          # a parameter that had no gradient last time must still have none (torch skips it; so does the table)
            for p in g['params']:
                if id(p) in self.with_grad:
                    k += 1
                elif p.grad is not None and p.requires_grad:
                    return False
        if k != len(self.params):
            return False
        st = opt.state
        for p, s_, (m, v) in zip(self.params, self.states, self.moments):
            s2 = st.get(p)
            if s2 is not s_ or s2['exp_avg'] is not m or s2['exp_avg_sq'] is not v:
                return False
        run = self.runner
        if tuple(id(_tiles_of(p)) for p, _, _ in run.tiled) != self.tiles:      # (a flag put on / taken off a parameter since)
            return False
        b1, b2, eps = self.hyper
        run.launch(b1, b2, eps)
        run.count = self.count = self.count + 1
        torch._foreach_add_(self.steps, 1)      # torch's bookkeeping: the per-parameter step counters
        return True


def adam_step(self, closure=None):
    """``torch.optim.Adam.step`` as ONE launch over every parameter (torch's foreach form: ~8 launches per parameter GROUP, 80 per step
    for the reference's ten groups: 0.84 ms of GPU time at config #1) when the optimizer is what the reference builds
    (gaussian_splatting.py:443-460: plain Adam, ``eps=1e-15``, one ``lr`` per group): dense fp32 parameters on one HIP device, every
    parameter that has a gradient with an initialised state (one without a gradient is skipped, as torch skips it; parameters of
    different step counts -- a later stage's groups -- take one launch per count), no amsgrad / weight decay / maximize /
    capturable / differentiable.  Anything else -- and the very first step, which creates the state -- is torch's own ``step``."""
    groups = self.param_groups
    g0 = groups[0] if groups else None
    plan = _adam_plans.get(self)
    if plan is not None and closure is None and plan.replay(self):
        calls['adam_fused'] += 1
        return None
    ok = closure is None and g0 is not None and type(self) is torch.optim.Adam
    buckets, dev = {}, None      # step count -> ([(param, grad, exp_avg, exp_avg_sq)], [lr])
    if ok:
        for g in groups:
            if (g.get('amsgrad') or g.get('weight_decay', 0) != 0 or g.get('maximize') or g.get('capturable') or g.get('differentiable')
                    or g['betas'] != g0['betas'] or g['eps'] != g0['eps'] or torch.is_tensor(g['lr'])):
                ok = False
                break
            for p in g['params']:
                if not p.requires_grad:
                    continue
                st = self.state.get(p)
                gr = p.grad
                if gr is None:      # (torch skips it too: no update, its step counter stays)
                    continue
                if (not st or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or gr.is_sparse
                        or not gr.is_contiguous() or gr.dtype != torch.float32 or (dev is not None and p.device != dev)):
                    ok = False
                    break
                c = st['step']
                c = float(c) if not torch.is_tensor(c) else (float(c) if not c.is_cuda else None)   # (a CPU scalar tensor: torch's default)
                if c is None:
                    ok = False
                    break
                dev = p.device
                ent, lrs = buckets.setdefault(c, ([], []))
                ent.append((p, gr, st['exp_avg'], st['exp_avg_sq']))
                lrs.append(float(g['lr']))
            if not ok:
                break
    # parameters that joined later (a new stage's groups, sk_gs.py) are younger than the others: one launch per step count, with its
    # own bias corrections; more than a few different ages is not the reference's optimizer any more
    if not ok or not buckets or len(buckets) > 4:
        calls['adam_reference'] += 1
        return _originals['adam'](self, closure)
    runners = _adam_runners.get(self)
    if runners is None:
        runners = _adam_runners[self] = {}
    newest = max(buckets)
    with torch.no_grad():
        for count in sorted(buckets, reverse=True):
            ent, lrs = buckets[count]
            age = newest - count          # (constant from step to step: the runner of a bucket keeps its table and its counter)
            run = runners.get(age)
            if run is None or run.dev != dev:
                run = runners[age] = _AdamRunner(dev)
            run.step(ent, lrs, float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']), int(count))
            for p, _, _, _ in ent:      # torch's bookkeeping: the per-parameter step counters (CPU scalars)
                self.state[p]['step'] += 1
    if len(buckets) == 1:   # the common case from the second fused step on: remember what was launched (``_AdamPlan``)
        ent, lrs = buckets[newest]
        _adam_plans[self] = _AdamPlan(self, runners[0], ent, int(newest) + 1)
    else:
        _adam_plans.pop(self, None)
    calls['adam_fused'] += 1
    return None


# ------------------------------------------------------------------------------------------------ install / restore
def _fully_imported(name: str) -> bool:
    """the module has finished executing (its classes exist): what a patch needs"""
    m = sys.modules.get(name)
    spec = getattr(m, '__spec__', None)
    return m is not None and not getattr(spec, '_initializing', False)


_WATCHED = ('networks.sk_gs', 'networks.losses.ssim', 'networks.losses.image_loss', 'networks.renderer.gaussian_render_origin')


class _PostImportPatcher:
    """``install_reference_hooks(accelerate=True)``: a meta-path finder that lets the normal machinery find the reference's modules and,
    once one of the three modules the fast paths live in has finished executing, applies the patches that have become possible
    (``accelerate_reference(strict=False)``) -- the rasterizer adapter right after its own module, i.e. before
    ``networks/gaussian_splatting.py`` binds its name (:34), so the model stores the typed-rotation wrapper without further ado"""

    def __init__(self, **kwargs):
        self.kwargs, self.busy = kwargs, False

    def find_spec(self, name, path=None, target=None):
        if name not in _WATCHED or self.busy:
            return None
        import importlib.util
        self.busy = True
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        finally:
            self.busy = False
        if spec is None or spec.loader is None or not hasattr(spec.loader, 'exec_module'):
            return spec
        loader, patcher = spec.loader, self

        class _Loader:
            def create_module(self, spec_):
                return loader.create_module(spec_) if hasattr(loader, 'create_module') else None

            def exec_module(self, module):
                loader.exec_module(module)
                spec_ = getattr(module, '__spec__', None)
                if spec_ is not None:
                    spec_._initializing = False     # (finished as far as the patches are concerned)
                accelerate_reference(strict=False, **patcher.kwargs)

            def __getattr__(self, item):
                return getattr(loader, item)
        spec.loader = _Loader()
        return spec


def install_post_import_patcher(**kwargs):
    """see ``_PostImportPatcher``; idempotent.  ``torch.optim.Adam.step`` (not part of the reference) is patched at once."""
    if not any(isinstance(f, _PostImportPatcher) for f in sys.meta_path):
        sys.meta_path.insert(0, _PostImportPatcher(**kwargs))
    return accelerate_reference(strict=False, **kwargs)



def accelerate_reference(ssim: bool = True, kinematic_chain: bool = True, networks: bool = True, lbs_weights: bool = True,
                         adam: bool = True, swizzle: bool = True, strict: bool = True, fused_render: bool = True) -> list:
    """Patch the methods on the reference's classes (the modules must be imported already; ``strict=False``: patch what IS imported,
    skip the rest -- what the post-import hook of ``install_reference_hooks(accelerate=True)`` calls as the modules arrive).  Returns what
    was patched."""
    done = []
    if not strict:
        sk, ss = _fully_imported('networks.sk_gs'), _fully_imported('networks.losses.ssim')
        networks, kinematic_chain, lbs_weights, ssim = networks and sk, kinematic_chain and sk, lbs_weights and sk, ssim and ss
        fused_render = fused_render and sk and ss and _fully_imported('networks.losses.image_loss')
    if fused_render:
        # the whole per-view step behind SkeletonGaussianSplatting.render + the two image terms (sk_gs_amd/reference_fused.py): render
        # falls back to the reference's own method -- and with it to the per-method fast paths below -- whenever its conditions fail
        from sk_gs_amd import reference_fused as rf
        mod, il = sys.modules.get('networks.sk_gs'), sys.modules.get('networks.losses.image_loss')
        if mod is None or il is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.sk_gs / networks.losses.image_loss are not loaded)")
        if 'render' not in _originals:
            _originals['render'] = mod.SkeletonGaussianSplatting.render
            mod.SkeletonGaussianSplatting.render = rf.render
        if 'image_loss' not in _originals:
            _originals['image_loss'] = il.ImageLoss.forward
            il.ImageLoss.forward = rf.image_loss_forward
        done += ['networks.sk_gs.SkeletonGaussianSplatting.render', 'networks.losses.image_loss.ImageLoss.forward']
    if networks:
        mod = sys.modules.get('networks.sk_gs')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.sk_gs is not loaded)")
        if 'sk_net' not in _originals:
            _originals['sk_net'] = mod.SimpleDeformationNetwork.forward
            mod.SimpleDeformationNetwork.forward = simple_deform_forward
        if 'sp_net' not in _originals:
            _originals['sp_net'] = mod.DeformNetwork.forward
            mod.DeformNetwork.forward = deform_network_forward
        done += ['networks.sk_gs.SimpleDeformationNetwork.forward', 'networks.sk_gs.DeformNetwork.forward']
    if lbs_weights:
        mod = sys.modules.get('networks.sk_gs')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.sk_gs is not loaded)")
        if 'w_sparse' not in _originals:   # the two regularisers the shipped stage-sp configuration runs on those weights every iteration
            _originals['w_sparse'], _originals['w_smooth'] = mod.SkeletonGaussianSplatting.loss_weight_sparsity, mod.SkeletonGaussianSplatting.loss_weight_smooth
            mod.SkeletonGaussianSplatting.loss_weight_sparsity = loss_weight_sparsity
            mod.SkeletonGaussianSplatting.loss_weight_smooth = loss_weight_smooth
        done += ['networks.sk_gs.SkeletonGaussianSplatting.loss_weight_sparsity', 'networks.sk_gs.SkeletonGaussianSplatting.loss_weight_smooth']
        if 'lbs_weight' not in _originals:
            _originals['lbs_weight'] = mod.SkeletonGaussianSplatting.calc_LBS_weight
            mod.SkeletonGaussianSplatting.calc_LBS_weight = calc_LBS_weight
        done.append('networks.sk_gs.SkeletonGaussianSplatting.calc_LBS_weight')
    if swizzle:
        mod = sys.modules.get('networks.renderer.gaussian_render_origin')
        if mod is not None and hasattr(mod, 'render_gs_offical') and (strict or _fully_imported('networks.renderer.gaussian_render_origin')):
            # (the adapter exists only where the upstream rasterizer package -- here: the stand-in -- could be imported)
            if 'render_adapter' not in _originals:
                _originals['render_adapter'] = mod.render_gs_offical
                mod.render_gs_offical = render_gs_offical
                user = sys.modules.get('networks.gaussian_splatting')     # (binds the name at import, gaussian_splatting.py:34,129)
                if user is not None and getattr(user, 'render_gs_offical', None) is _originals['render_adapter']:
                    user.render_gs_offical = render_gs_offical
            done.append('networks.renderer.gaussian_render_origin.render_gs_offical')
    if adam:
        if 'adam' not in _originals:
            _originals['adam'] = torch.optim.Adam.step
            step = adam_step
            if getattr(_originals['adam'], 'hooked', False):
                # torch wrapped the class's step once, when the first optimizer was built (Optimizer._patch_step_function: the wrapper
                # runs the registered step pre / post hooks and the profiler range): an optimizer that exists already must keep them
                step = torch.optim.Optimizer.profile_hook_step(adam_step)
                step.hooked = True
            torch.optim.Adam.step = step
        done.append('torch.optim.Adam.step')
    if ssim:
        mod = sys.modules.get('networks.losses.ssim')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.losses.ssim is not loaded)")
        if 'ssim' not in _originals:
            _originals['ssim'] = mod.SSIM_Loss.forward
            mod.SSIM_Loss.forward = ssim_loss_forward
        done.append('networks.losses.ssim.SSIM_Loss.forward')
    if kinematic_chain:
        mod = sys.modules.get('networks.sk_gs')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.sk_gs is not loaded)")
        if 'kinematic' not in _originals:
            _originals['kinematic'] = mod.SkeletonGaussianSplatting.kinematic
            mod.SkeletonGaussianSplatting.kinematic = kinematic
        done.append('networks.sk_gs.SkeletonGaussianSplatting.kinematic')
    return done


def restore_reference():
    """put the reference's own methods back"""
    if 'render' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.render = _originals.pop('render')
        from sk_gs_amd import reference_fused as rf
        for ref in list(rf._routes.keys()):     # the heads the fused route re-homed get storage of their own again
            if hasattr(ref, 'sk_deform_net'):
                rf.unhome_heads(ref.sk_deform_net)
            if getattr(getattr(ref, 'sp_W', None), '_skgs_logit_tiles', None) is not None:
                del ref.sp_W._skgs_logit_tiles
        rf._routes.clear()
    if 'image_loss' in _originals and 'networks.losses.image_loss' in sys.modules:
        sys.modules['networks.losses.image_loss'].ImageLoss.forward = _originals.pop('image_loss')
    if 'ssim' in _originals and 'networks.losses.ssim' in sys.modules:
        sys.modules['networks.losses.ssim'].SSIM_Loss.forward = _originals.pop('ssim')
    if 'kinematic' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.kinematic = _originals.pop('kinematic')
    if 'sk_net' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SimpleDeformationNetwork.forward = _originals.pop('sk_net')
    if 'sp_net' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].DeformNetwork.forward = _originals.pop('sp_net')
    if 'adam' in _originals:
        torch.optim.Adam.step = _originals.pop('adam')
        _adam_plans.clear()
    if 'render_adapter' in _originals:
        orig = _originals.pop('render_adapter')
        for name in ('networks.renderer.gaussian_render_origin', 'networks.gaussian_splatting'):
            m = sys.modules.get(name)
            if m is not None and getattr(m, 'render_gs_offical', None) is render_gs_offical:
                m.render_gs_offical = orig
    if 'lbs_weight' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.calc_LBS_weight = _originals.pop('lbs_weight')
    if 'w_sparse' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.loss_weight_sparsity = _originals.pop('w_sparse')
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.loss_weight_smooth = _originals.pop('w_smooth')
