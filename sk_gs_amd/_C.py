"""Host binding of ``libskgs_hip.so`` (C ABI in ``include/skgs.h``) under the reference's extension surface.

The reference resolves its CUDA ops by name through ``my_ext._C.get_C_function(name)`` (my_ext/_C/__init__.py:17-48)
and calls them with torch tensors.  This module offers the same names with the same positional arguments and the same
returned tuples, implemented by passing raw device pointers + the current HIP stream to the C ABI.  PyTorch is used
for device memory and streams only.

There is NO fallback: if the shared library is missing (or a tensor is not on a HIP device) the call raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional, Tuple

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
# SKGS_HIP_LIB: another build of the same library (kernel experiments, tools/); there is no non-HIP implementation
_LIB_PATH = os.environ.get('SKGS_HIP_LIB') or os.path.join(_HERE, 'libskgs_hip.so')
_lib = None
_lock = threading.Lock()


class SkgsError(RuntimeError):
    pass


class _KnnDeformJob(C.Structure):
    """include/skgs.h: skgs_knn_deform_job"""
    _fields_ = [('M', C.c_int32), ('K', C.c_int32)] + [(n, C.c_void_p) for n in (
        'points', 'joints', 'sp_W', 'bone_T', 'bone_drot', 'bone_dscale', 'xyz', 'log_scale', 'rot', 'opacity_logit', 'out_idx',
        'out_weights', 'means', 'scales', 'rotations', 'opacity')] + [('largest', C.c_int32)]


class _RasterInputs(C.Structure):
    _fields_ = [
        ('P', C.c_int32), ('sh_degree', C.c_int32), ('sh_coeffs', C.c_int32), ('E', C.c_int32),
        ('image_height', C.c_int32), ('image_width', C.c_int32),
        ('tanfovx', C.c_float), ('tanfovy', C.c_float), ('scale_modifier', C.c_float),
        ('prefiltered', C.c_int32), ('debug', C.c_int32), ('colmap', C.c_int32),
        ('viewmatrix', C.c_void_p), ('projmatrix', C.c_void_p), ('campos', C.c_void_p),
        ('means3D', C.c_void_p), ('opacity', C.c_void_p), ('sh', C.c_void_p), ('scales', C.c_void_p),
        ('rotations', C.c_void_p), ('extras', C.c_void_p), ('colors_precomp', C.c_void_p),
        ('cov3D_precomp', C.c_void_p), ('sh_rest', C.c_void_p), ('background', C.c_void_p),
        ('tile_bucket_capacity', C.c_int32), ('tanfov_device', C.c_void_p),
        ('host_status_words', C.c_int32), ('longest_list_hint', C.c_int32), ('live_count', C.c_void_p),
        ('deform_job', C.POINTER(_KnnDeformJob)), ('tiles_per_gaussian_hint', C.c_int32),
    ]


class _RasterBuffers(C.Structure):
    _fields_ = [('geom', C.c_void_p), ('geom_bytes', C.c_size_t), ('binning', C.c_void_p),
                ('binning_bytes', C.c_size_t), ('img', C.c_void_p), ('img_bytes', C.c_size_t)]


class _Status(C.Structure):
    _fields_ = [('num_rendered', C.c_int32), ('overflow', C.c_int32), ('max_tile_count', C.c_int32),
                ('overflow_events', C.c_int32)]


class _RasterGrads(C.Structure):
    _fields_ = [
        ('dL_dout_color', C.c_void_p), ('dL_dout_opacity', C.c_void_p), ('dL_dout_extra', C.c_void_p),
        ('grad_means2D_in', C.c_void_p), ('grad_conic_in', C.c_void_p), ('grad_opacity_in', C.c_void_p),
        ('dL_dmeans2D', C.c_void_p), ('dL_dconic', C.c_void_p), ('dL_dcolors', C.c_void_p),
        ('dL_dopacity', C.c_void_p), ('dL_dmeans3D', C.c_void_p), ('dL_dcov3D', C.c_void_p), ('dL_dsh', C.c_void_p),
        ('dL_dscales', C.c_void_p), ('dL_drotations', C.c_void_p), ('dL_dextras', C.c_void_p),
        ('dL_dsh_rest', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
        ('workspace_is_zero', C.c_int32), ('dL_dsh_factors', C.c_void_p),
        ('stat_xyz_gradient_accum', C.c_void_p), ('stat_denom', C.c_void_p), ('stat_max_radii2D', C.c_void_p),
        ('stat_grad_multiplier', C.c_float), ('deform_backward_job', C.c_void_p), ('sp_skinning_job', C.c_void_p),
    ]


class _DeformInputs(C.Structure):
    _fields_ = [
        ('P', C.c_int32), ('K', C.c_int32), ('M', C.c_int32),
        ('points', C.c_void_p), ('weights', C.c_void_p), ('indices', C.c_void_p), ('bone_T', C.c_void_p),
        ('bone_drot', C.c_void_p), ('bone_dscale', C.c_void_p), ('xyz', C.c_void_p), ('log_scale', C.c_void_p),
        ('rot', C.c_void_p), ('opacity_logit', C.c_void_p), ('live_count', C.c_void_p), ('largest', C.c_int32),
    ]


class _DeformBackwardJob(C.Structure):
    """include/skgs.h: skgs_deform_backward_job"""
    _fields_ = [('in_', C.POINTER(_DeformInputs))] + [(n, C.c_void_p) for n in (
        'g_bone_T', 'g_bone_drot', 'g_bone_dscale', 'g_xyz', 'g_log_scale', 'g_rot', 'g_opacity_logit', 'g_sp_W', 'g_logits',
        'workspace')] + [('workspace_bytes', C.c_size_t)]


class _SpSkinningJob(C.Structure):
    """include/skgs.h: skgs_sp_skinning_job"""
    _fields_ = ([('in_', C.POINTER(_DeformInputs)), ('F', C.c_int32)] +
                [(n, C.c_void_p) for n in ('feature', 'sp_feature', 'sp_radius_raw', 'sp_weight_raw')] +
                [('temperature', C.c_float), ('logit_weighting', C.c_int32), ('nn_dist', C.c_void_p)] +
                [(n, C.c_void_p) for n in ('g_weights', 'g_xyz', 'g_log_scale', 'g_rot', 'g_opacity_logit', 'g_feature', 'g_bone_T',
                                           'g_bone_drot', 'g_bone_dscale', 'g_sp_feature', 'g_sp_radius', 'g_sp_weight', 'pairs')] +
                [('pairs_bytes', C.c_size_t), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('g_weights_extra', C.c_void_p)])


EXPORTED_SYMBOLS = [
    'skgs_geom_buffer_bytes', 'skgs_img_buffer_bytes', 'skgs_binning_buffer_bytes', 'skgs_binning_capacity',
    'skgs_rasterize_forward_stage1', 'skgs_rasterize_forward_stage2', 'skgs_rasterize_forward', 'skgs_read_status',
    'skgs_backward_workspace_bytes', 'skgs_rasterize_backward', 'skgs_rasterize_extra_forward',
    'skgs_rasterize_extra_backward', 'skgs_topk_weights', 'skgs_render_census', 'skgs_set_tile_order', 'skgs_knn_dist_weights_forward', 'skgs_knn_dist_weights_backward',
    'skgs_knn_dist_weights_workspace_bytes', 'skgs_mark_visible', 'skgs_lbs_deform_forward',
    'skgs_lbs_deform_backward', 'skgs_lbs_deform_backward_workspace_bytes', 'skgs_knn_bones',
    'skgs_lbs_weights_forward', 'skgs_lbs_weights_backward', 'skgs_last_error', 'skgs_version',
]


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen libskgs_hip.so (once). Raises SkgsError with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise SkgsError(
                f'{_LIB_PATH} not found: the HIP extension is not built. Run `python -c "import __graft_entry__ as g; '
                f'g.build()"` or `make -C sk_gs_amd/csrc`. There is no CPU / PyTorch fallback for this path.')
        lib = C.CDLL(_LIB_PATH)
        for name in ('skgs_geom_buffer_bytes', 'skgs_img_buffer_bytes', 'skgs_binning_buffer_bytes',
                     'skgs_backward_workspace_bytes', 'skgs_lbs_deform_backward_workspace_bytes',
                     'skgs_image_loss_workspace_bytes'):
            getattr(lib, name).restype = C.c_size_t
        lib.skgs_binning_capacity.restype = C.c_int64
        lib.skgs_binning_capacity.argtypes = [C.c_size_t]
        lib.skgs_binning_buffer_bytes.argtypes = [C.c_int64]
        lib.skgs_last_error.restype = C.c_char_p
        if os.environ.get('SKGS_TILE_ORDER') is not None:  # A/B timing of the blend kernels' tile order (0: raster order)
            lib.skgs_set_tile_order(C.c_int(int(os.environ['SKGS_TILE_ORDER'])))
        _lib = lib
    return _lib


_ops = None


def _torch_ops():
    """the C++ tensor-level entry points (csrc/torch_ops.cpp -> _skgs_torch.so): the same marshalling as the ctypes code
    below, compiled.  None when the module is not built, when SKGS_TORCH_OPS=0, or when SKGS_HIP_LIB points at another
    build of the kernels (the module is linked against the in-tree libskgs_hip.so)."""
    global _ops
    if _ops is None:
        _ops = False
        if os.environ.get('SKGS_TORCH_OPS', '1') != '0' and not os.environ.get('SKGS_HIP_LIB'):
            load_library()  # the kernels' library first: the module resolves its symbols against the loaded instance
            try:
                from sk_gs_amd import _skgs_torch
                _ops = _skgs_torch
            except ImportError:
                pass
    return _ops or None


def _check(rc: int):
    if rc != 0:
        raise SkgsError(load_library().skgs_last_error().decode())


def _ptr(t: Optional[Tensor]):
    """device pointer or NULL for None / empty tensors (the reference's null-data_ptr convention)"""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _f32c(t: Optional[Tensor], device=None) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype is torch.float32 and t.is_contiguous() and (device is None or t.device == device):
        return t  # the common case on the operator path: nothing to convert
    if t.numel() == 0:
        return t
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_gpu(t: Tensor, name: str):
    if not t.is_cuda:
        raise SkgsError(f'{name} must live on a HIP device (got {t.device}); sk_gs_amd has no CPU path')


_NULLCTX = contextlib.nullcontext()


def _on_device(dev):
    """device guard that is a no-op when `dev` already is the current device: switching devices (hipSetDevice)
    inside a stream capture is not safe on ROCm, and the common single-device-per-process case never needs it"""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NULLCTX
    return torch.cuda.device(dev)


def _stream() -> C.c_void_p:
    """the current HIP stream of the current device as a raw handle (torch.cuda.current_stream() builds a Stream object:
    ~7 us per call, twice per render on the operator path)"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


# ----------------------------------------------------------------------------------------------- configuration
class _Config:
    #: True  -> reference behaviour: one host sync per forward to size the binning buffer exactly and return R
    #: False -> no sync: binning capacity is predicted from the previous call (x growth) and checked lazily
    sync_num_rendered: bool = True
    capacity_growth: float = 1.5
    min_capacity: int = 1 << 16


config = _Config()
_capacity_hint = {}  # (P, W, H) -> last capacity
_bucket_hint = {}    # (P, W, H) -> slots per tile for the bucket layout of the tile lists (sync-free forward only)
_zero_ws = {}        # (device index, P) -> backward scratch that every backward hands back all zero
_pinned = {}


def _pinned_i32(device, n=4) -> Tensor:
    key = (device.index, n)
    t = _pinned.get(key)
    if t is None:
        t = torch.zeros(n, dtype=torch.int32).pin_memory()
        _pinned[key] = t
    return t


def set_pixels_per_lane(ppl: int):
    """tuning knob of the blend kernels: 1, 2 or 4 pixels per lane; 0 = heuristic"""
    load_library().skgs_set_pixels_per_lane(C.c_int(ppl))


def profile_kernels() -> dict:
    """name -> kernel id of the kernels the library can time with HIP events"""
    lib = load_library()
    lib.skgs_profile_kernel_name.restype = C.c_char_p
    return {lib.skgs_profile_kernel_name(C.c_int(i)).decode(): i for i in range(lib.skgs_profile_kernel_count())}


def profile_enable(names=None):
    """enable HIP-event timing for the named kernels (None = all, [] = off)"""
    ids = profile_kernels()
    mask = 0
    for n in (ids.keys() if names is None else names):
        mask |= 1 << ids[n]
    load_library().skgs_profile_enable(C.c_uint32(mask))


def profile_collect() -> dict:
    """name -> (total_ms, launches) since the last collect (synchronises on the recorded events)"""
    lib = load_library()
    out = {}
    for name, kid in profile_kernels().items():
        ms, n = C.c_double(0), C.c_int32(0)
        _check(lib.skgs_profile_collect(C.c_int(kid), C.byref(ms), C.byref(n)))
        if n.value:
            out[name] = (ms.value, n.value)
    return out


def densify_stats(radii: Tensor, grad_means2D: Tensor, xyz_gradient_accum: Tensor, denom: Tensor, max_radii2D: Tensor,
                  grad_multiplier: float = 1.0):
    """In-place densification statistics of one view (``add_densification_stats`` + the ``max_radii2D`` update,
    networks/gaussian_splatting.py:503-513, networks/sk_gs.py:1990-1997) in one launch.  ``grad_multiplier`` undoes a
    pre-scaled backward (view-parallel training seeds it with 1 / world): the statistic is defined on the unscaled
    screen-space gradient."""
    lib = load_library()
    _require_gpu(radii, 'radii')
    P = radii.shape[0]
    for t, n in ((radii, P), (grad_means2D, 3 * P), (xyz_gradient_accum, P), (denom, P), (max_radii2D, P)):
        if not (t.is_cuda and t.is_contiguous() and t.numel() == n):
            raise SkgsError('densify_stats: tensors must be contiguous device tensors of P rows')
    if radii.dtype != torch.int32 or any(t.dtype != torch.float32 for t in (grad_means2D, xyz_gradient_accum, denom, max_radii2D)):
        raise SkgsError('densify_stats: radii int32, the others float32')
    _check(lib.skgs_densify_stats(C.c_int32(P), C.c_void_p(_ptr(radii)), C.c_void_p(_ptr(grad_means2D)),
                                  C.c_float(float(grad_multiplier)), C.c_void_p(_ptr(xyz_gradient_accum)), C.c_void_p(_ptr(denom)),
                                  C.c_void_p(_ptr(max_radii2D)), _stream()))


def set_tile_order(mode: int):
    """1 (default): the blend kernels walk groups of 8 tiles heaviest first; 0: raster order (``skgs_set_tile_order``)"""
    load_library().skgs_set_tile_order(C.c_int(int(mode)))


def set_strict_math(on: bool):
    """parity-test switch: blend kernels built without FMA contraction, in the oracle's operation order, with the
    reproducible double-arithmetic exp (bit-comparable with the oracle's exp_mode=1). Slower; never used by bench."""
    load_library().skgs_set_strict_math(C.c_int(int(bool(on))))


_size_cache = {}


def _buffer_bytes(lib, kind: str, *dims) -> int:
    """skgs_*_buffer_bytes, memoised (three ctypes calls per forward otherwise)"""
    key = (kind,) + dims
    v = _size_cache.get(key)
    if v is None:
        if kind == 'geom':
            v = lib.skgs_geom_buffer_bytes(C.c_int32(dims[0]))
        elif kind == 'img':
            v = lib.skgs_img_buffer_bytes(C.c_int32(dims[0]), C.c_int32(dims[1]))
        elif kind == 'bwd_ws':
            v = lib.skgs_backward_workspace_bytes(C.c_int32(dims[0]))
        else:
            v = lib.skgs_binning_buffer_bytes(C.c_int64(dims[0]))
        _size_cache[key] = v = int(v)
    return v


def _make_inputs(H, W, tanfovx, tanfovy, degree, scale_modifier, prefiltered, debug, colmap, viewmatrix, projmatrix,
                 campos, means3D, opacity, sh, scales, rotations, extras, colors, cov3D_precomp):
    dev = means3D.device
    keep = []

    def prep(t):
        t = _f32c(t, dev)
        keep.append(t)
        return t

    means3D = prep(means3D)
    P = means3D.shape[0]
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise SkgsError('means3D must have dimensions (num_points, 3)')
    sh = prep(sh)
    M = sh.shape[1] if (sh is not None and sh.numel() > 0) else 0
    extras = prep(extras)
    E = extras.shape[-1] if (extras is not None and extras.numel() > 0) else 0
    a = _RasterInputs()
    a.P, a.sh_degree, a.sh_coeffs, a.E = P, int(degree), M, E
    a.image_height, a.image_width = int(H), int(W)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
    a.prefiltered, a.debug, a.colmap = int(bool(prefiltered)), int(bool(debug)), int(bool(colmap))
    a.viewmatrix, a.projmatrix, a.campos = _ptr(prep(viewmatrix)), _ptr(prep(projmatrix)), _ptr(prep(campos))
    a.means3D, a.opacity, a.sh = _ptr(means3D), _ptr(prep(opacity)), _ptr(sh)
    a.scales, a.rotations, a.extras = _ptr(prep(scales)), _ptr(prep(rotations)), _ptr(extras)
    a.colors_precomp, a.cov3D_precomp = _ptr(prep(colors)), _ptr(prep(cov3D_precomp))
    return a, keep, P, M, E


def _buffers(geom: Tensor, binning: Tensor, img: Tensor) -> _RasterBuffers:
    b = _RasterBuffers()
    b.geom, b.geom_bytes = geom.data_ptr(), geom.numel()
    b.binning, b.binning_bytes = (binning.data_ptr() if binning.numel() else None), binning.numel()
    b.img, b.img_bytes = img.data_ptr(), img.numel()
    return b


# =============================================================================================== rasterize_gaussians
def rasterize_gaussians(image_height: int, image_width: int, tanfovx: float, tanfovy: float, degree: int,
                        scale_modifier: float, prefiltered: bool, debug: bool, colmap: bool, viewmatrix: Tensor,
                        projmatrix: Tensor, campos: Tensor, means3D: Tensor, opacity: Tensor, sh: Tensor,
                        scales: Tensor, rotations: Tensor, extras: Optional[Tensor], colors: Tensor,
                        cov3D_precomp: Tensor):
    """Drop-in for ``_C.rasterize_gaussians`` (gaussian_rasterizer_forward.cu:260-317).

    Returns ``(num_rendered, color[3,H,W], opacity[H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer,
    out_extras[E,H,W] | None)``.
    """
    lib = load_library()
    _require_gpu(means3D, 'means3D')
    dev = means3D.device
    H, W = int(image_height), int(image_width)
    ops = None if config.sync_num_rendered else _torch_ops()
    if ops is not None and means3D.ndim == 2 and means3D.shape[0] > 0 and not (
            sh is None or scales is None or rotations is None or colors is None or cov3D_precomp is None):
        P = means3D.shape[0]
        key = (P, W, H)
        bucket = _bucket_hint.get(key, 0)
        cap = ((W + 15) // 16) * ((H + 15) // 16) * bucket if bucket else _capacity_hint.get(
            key, max(config.min_capacity, 8 * P))
        with _on_device(dev):
            try:
                out_color, out_opacity, radii, geom, binning, img, out_extras = ops.rasterize_forward(
                    H, W, float(tanfovx), float(tanfovy), int(degree), float(scale_modifier), bool(prefiltered), bool(debug),
                    bool(colmap), viewmatrix, projmatrix, campos, means3D, opacity, sh, scales, rotations, extras, colors,
                    cov3D_precomp, _buffer_bytes(lib, 'geom', P), _buffer_bytes(lib, 'img', W, H),
                    _buffer_bytes(lib, 'binning', int(cap)), bucket,
                    torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
            except RuntimeError as e:
                raise SkgsError(str(e)) from None
        return -1, out_color, out_opacity, radii, geom, binning, img, out_extras  # R unknown without a sync: read_status()
    with _on_device(dev):
        a, keep, P, M, E = _make_inputs(H, W, tanfovx, tanfovy, degree, scale_modifier, prefiltered, debug, colmap,
                                        viewmatrix, projmatrix, campos, means3D, opacity, sh, scales, rotations,
                                        extras, colors, cov3D_precomp)
        f32 = dict(dtype=torch.float32, device=dev)
        out_color = torch.empty((3, H, W), **f32)
        out_opacity = torch.empty((H, W), **f32)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        out_extras = torch.empty((E, H, W), **f32) if extras is not None and E > 0 else None
        geom = torch.empty((_buffer_bytes(lib, 'geom', P),), dtype=torch.uint8, device=dev)
        img = torch.empty((_buffer_bytes(lib, 'img', W, H),), dtype=torch.uint8, device=dev)
        if P == 0:
            out_color.zero_(), out_opacity.zero_()
            if out_extras is not None:
                out_extras.zero_()
            binning = torch.empty((0,), dtype=torch.uint8, device=dev)
            geom[:256].zero_()
            return 0, out_color, out_opacity, radii, geom, binning, img, out_extras
        stream = _stream()
        if config.sync_num_rendered:
            host_r = _pinned_i32(dev)
            bufs = _buffers(geom, torch.empty((0,), dtype=torch.uint8, device=dev), img)
            a.host_status_words = 3  # R, overflow flag, longest tile list: stage 2 skips the sort launches no list needs
            _check(lib.skgs_rasterize_forward_stage1(C.byref(a), C.byref(bufs), C.c_void_p(radii.data_ptr()),
                                                     C.c_void_p(host_r.data_ptr()), stream))
            torch.cuda.current_stream().synchronize()
            num_rendered = int(host_r[0])
            a.longest_list_hint = max(int(host_r[2]), 1)
            binning = torch.empty((lib.skgs_binning_buffer_bytes(C.c_int64(num_rendered)),), dtype=torch.uint8,
                                  device=dev)
            bufs = _buffers(geom, binning, img)
            _check(lib.skgs_rasterize_forward_stage2(C.byref(a), C.byref(bufs), C.c_void_p(out_color.data_ptr()),
                                                     C.c_void_p(out_opacity.data_ptr()),
                                                     C.c_void_p(_ptr(out_extras)), stream))
        else:
            key = (P, W, H)
            cap = _capacity_hint.get(key, max(config.min_capacity, 8 * P))
            bucket = _bucket_hint.get(key, 0)
            if bucket:  # fixed slots per tile: preprocess -> scatter -> sort -> blend, four launches
                a.tile_bucket_capacity = bucket
                cap = ((W + 15) // 16) * ((H + 15) // 16) * bucket
            binning = torch.empty((_buffer_bytes(lib, 'binning', int(cap)),), dtype=torch.uint8, device=dev)
            bufs = _buffers(geom, binning, img)
            _check(lib.skgs_rasterize_forward(C.byref(a), C.byref(bufs), C.c_void_p(radii.data_ptr()),
                                              C.c_void_p(out_color.data_ptr()), C.c_void_p(out_opacity.data_ptr()),
                                              C.c_void_p(_ptr(out_extras)), None, stream))
            num_rendered = -1  # unknown without a sync: see read_status()
    return num_rendered, out_color, out_opacity, radii, geom, binning, img, out_extras


def read_status(geomBuffer: Tensor) -> dict:
    """(synchronising) status words of a forward: num_rendered, overflow flag, longest tile list"""
    hdr = geomBuffer[:16].cpu().view(torch.int32)
    return dict(num_rendered=int(hdr[0]), overflow=int(hdr[1]), max_tile_count=int(hdr[2]),
                overflow_events=int(hdr[3]))  # sticky counter: meaningful when the caller zeroed the header once


def unpack_buffers(W: int, H: int, P: int, geomBuffer: Tensor, binningBuffer: Tensor, imgBuffer: Tensor) -> dict:
    """Diagnostic view of the opaque buffers (layout: csrc/skgs_common.h). Used by tests and tools only."""
    def a256(x):
        return (x + 255) & ~255
    T = ((W + 15) // 16) * ((H + 15) // 16)
    hdr = geomBuffer[:16].view(torch.int32)
    recs = geomBuffer[256:256 + P * 48].view(torch.float32).view(P, 12)
    o = 0
    n_contrib = imgBuffer[o:o + W * H * 4].view(torch.int32).view(H, W)
    o += a256(W * H * 4)
    tile_counts = imgBuffer[o:o + T * 4].view(torch.int32)
    o += a256(T * 4)
    tile_offsets = imgBuffer[o:o + (T + 1) * 4].view(torch.int32)
    o += a256((T + 1) * 4) + 4 * a256(T * 4)  # cursors, tile_begin, tile_end, worklist
    G = (T + 7) // 8
    group_order = imgBuffer[o:o + G * 4].view(torch.int32)
    cap = int(load_library().skgs_binning_capacity(C.c_size_t(binningBuffer.numel())))
    keys = binningBuffer[:cap * 8].view(torch.int64)
    plo = a256(cap * 8)
    point_list = binningBuffer[plo:plo + cap * 4].view(torch.int32)
    return dict(num_rendered=hdr[0], overflow=hdr[1], max_tile_count=hdr[2], recs=recs, n_contrib=n_contrib,
                tile_counts=tile_counts, tile_offsets=tile_offsets, group_order=group_order, keys=keys, point_list=point_list,
                capacity=cap)


def update_capacity_hint(P: int, W: int, H: int, num_rendered: int, longest_list: int = 0):
    """sync-free forwards (``config.sync_num_rendered = False``) of this shape get a binning buffer for ``num_rendered`` x
    growth tile instances.  With ``longest_list`` (the longest tile list seen, ``read_status()['max_tile_count']``) they use
    the BUCKET layout instead -- every tile owns 1.5 x that many fixed slots, rounded up to 64: no counting and no scan
    launch, and no merge-sort launch when a wave sorts every bucket (what ``FusedViewStep(tile_bucket=...)`` does); a tile
    that outgrows its bucket raises the same device-side overflow flag as a binning buffer that is too small."""
    _capacity_hint[(P, W, H)] = max(config.min_capacity, int(num_rendered * config.capacity_growth) + 1024)
    if longest_list > 0:
        bucket = ((int(longest_list * 1.5) + 63) // 64) * 64
        if 512 < bucket and longest_list * 1.2 <= 512:
            bucket = 512
        _bucket_hint[(P, W, H)] = bucket
    else:
        _bucket_hint.pop((P, W, H), None)


# ====================================================================================== rasterize_gaussians_backward
def rasterize_gaussians_backward(scale_modifier: float, tanfovx: float, tanfovy: float, degree: int, debug: bool,
                                 colmap: bool, viewmatrix: Tensor, projmatrix: Tensor, campos: Tensor,
                                 means3D: Tensor, colors: Tensor, extras: Optional[Tensor], scales: Tensor,
                                 rotations: Tensor, cov3D_precomp: Tensor, sh: Tensor, R: int, radii: Tensor,
                                 out_opacity: Tensor, dL_dout_color: Tensor, dL_dout_opacity: Tensor,
                                 dL_dout_extra: Optional[Tensor], grad_means2D: Optional[Tensor],
                                 grad_conic: Optional[Tensor], grad_opacity: Optional[Tensor], geomBuffer: Tensor,
                                 binningBuffer: Tensor, imgBuffer: Tensor):
    """Drop-in for ``_C.rasterize_gaussians_backward`` (gaussian_rasterizer_backwrad.cu:200-261).

    Returns ``(dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
    dL_dscales[P,3], dL_drotations[P,4], dL_dextras[P,E] | None)``.
    """
    lib = load_library()
    _require_gpu(means3D, 'means3D')
    dev = means3D.device
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    ops = _torch_ops()
    if ops is not None and means3D.ndim == 2 and means3D.shape[0] > 0 and dL_dout_opacity is not None and not (
            sh is None or scales is None or rotations is None or colors is None or cov3D_precomp is None):
        P = means3D.shape[0]
        with _on_device(dev):
            stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
            capturing = torch.cuda.is_current_stream_capturing()
            ws_key = (dev.index, P, stream)  # (the same pool of all-zero scratch rows as the ctypes path below)
            ws = _zero_ws.pop(ws_key, None)
            if ws is None or capturing:
                ws = torch.zeros((_buffer_bytes(lib, 'bwd_ws', P),), dtype=torch.uint8, device=dev)
            try:
                out = ops.rasterize_backward(
                    float(scale_modifier), float(tanfovx), float(tanfovy), int(degree), bool(debug), bool(colmap), viewmatrix,
                    projmatrix, campos, means3D, colors, extras, scales, rotations, cov3D_precomp, sh, radii, out_opacity,
                    dL_dout_color, dL_dout_opacity, dL_dout_extra, grad_means2D, grad_conic, grad_opacity, geomBuffer,
                    binningBuffer, imgBuffer, ws, stream)
            except RuntimeError as e:
                raise SkgsError(str(e)) from None
            if not capturing and len(_zero_ws) < 8:
                _zero_ws[ws_key] = ws
        return out
    with _on_device(dev):
        # opacity is not an input of the backward: the blend kernels read it from the saved records
        dummy_op = means3D  # any non-null pointer satisfies the input check; never dereferenced in the backward
        a, keep, P, M, E = _make_inputs(H, W, tanfovx, tanfovy, degree, scale_modifier, False, debug, colmap,
                                        viewmatrix, projmatrix, campos, means3D, dummy_op, sh, scales, rotations,
                                        extras, colors, cov3D_precomp)
        f32 = dict(dtype=torch.float32, device=dev)
        use_extra = extras is not None and dL_dout_extra is not None and E > 0
        g = _RasterGrads()
        dL_dout_color = _f32c(dL_dout_color, dev)
        dL_dout_opacity = _f32c(dL_dout_opacity, dev)
        dL_dout_extra = _f32c(dL_dout_extra, dev) if use_extra else None
        gm_in, gc_in, go_in = _f32c(grad_means2D, dev), _f32c(grad_conic, dev), _f32c(grad_opacity, dev)
        out_opacity = _f32c(out_opacity, dev)
        radii = radii.contiguous()
        dL_dmeans2D = torch.empty((P, 3), **f32)
        dL_dcolors = torch.empty((P, 3), **f32)
        dL_dopacity = torch.empty((P, 1), **f32)
        dL_dmeans3D = torch.empty((P, 3), **f32)
        dL_dcov3D = torch.empty((P, 6), **f32)
        dL_dsh = torch.empty((P, M, 3), **f32)
        dL_dscales = torch.empty((P, 3), **f32)
        dL_drot = torch.empty((P, 4), **f32)
        dL_dextras = torch.empty((P, E), **f32) if use_extra else None
        if P == 0:
            return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot,
                    dL_dextras)
        # the scratch rows of the blend backward: pooled per (device, P) and handed back all zero by every backward
        # (skgs_raster_grads.workspace_is_zero) -- no 64-byte-per-Gaussian fill launch in front of each call
        ws_bytes = _buffer_bytes(lib, 'bwd_ws', P)
        ws_key = (dev.index, P, _stream().value)  # (per stream: nothing orders two streams' use of one scratch)
        ws = _zero_ws.pop(ws_key, None)  # (popped: a call that fails half-way must not leave a dirty scratch behind)
        if ws is None or torch.cuda.is_current_stream_capturing():
            ws = torch.zeros((ws_bytes,), dtype=torch.uint8, device=dev)
        g.workspace_is_zero = 1
        g.dL_dout_color, g.dL_dout_opacity, g.dL_dout_extra = _ptr(dL_dout_color), _ptr(dL_dout_opacity), _ptr(dL_dout_extra)
        g.grad_means2D_in, g.grad_conic_in, g.grad_opacity_in = _ptr(gm_in), _ptr(gc_in), _ptr(go_in)
        g.dL_dmeans2D, g.dL_dconic, g.dL_dcolors, g.dL_dopacity = _ptr(dL_dmeans2D), None, _ptr(dL_dcolors), _ptr(dL_dopacity)
        g.dL_dmeans3D, g.dL_dcov3D, g.dL_dsh = _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh)
        g.dL_dscales, g.dL_drotations, g.dL_dextras = _ptr(dL_dscales), _ptr(dL_drot), _ptr(dL_dextras)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws_bytes
        if not use_extra:
            a.extras, a.E = None, 0
        bufs = _buffers(geomBuffer, binningBuffer, imgBuffer)
        _check(lib.skgs_rasterize_backward(C.byref(a), C.byref(bufs), C.c_void_p(radii.data_ptr()),
                                           C.c_void_p(out_opacity.data_ptr()), C.byref(g), _stream()))
        if not torch.cuda.is_current_stream_capturing() and len(_zero_ws) < 8:
            _zero_ws[ws_key] = ws
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot, dL_dextras


# ======================================================================================================= extras / topk
def gaussian_rasterize_extra_forward(W: int, H: int, R: int, extra: Tensor, geomBuffer: Tensor,
                                     binningBuffer: Tensor, imgBuffer: Tensor) -> Tensor:
    """Drop-in for ``_C.gaussian_rasterize_extra_forward`` (gaussian_rasterizer_extra.cu:222-246).
    Returns a tensor labelled ``[W, H, E]`` whose memory is pixel-major ``[H*W, E]`` -- exactly like the reference."""
    lib = load_library()
    _require_gpu(extra, 'extras')
    if extra.ndim != 2:
        raise SkgsError('Error shape for extras')
    dev = extra.device
    with _on_device(dev):
        extra = _f32c(extra, dev)
        P, E = extra.shape
        out = torch.zeros((W, H, E), dtype=torch.float32, device=dev)
        if P == 0:
            return out
        bufs = _buffers(geomBuffer, binningBuffer, imgBuffer)
        _check(lib.skgs_rasterize_extra_forward(C.c_int32(W), C.c_int32(H), C.c_int32(P), C.c_int32(E),
                                                C.c_void_p(extra.data_ptr()), C.byref(bufs),
                                                C.c_void_p(out.data_ptr()), _stream()))
    return out


def gaussian_rasterize_extra_backward(W: int, H: int, R: int, extra: Tensor, out_opacity: Tensor,
                                      grad_pixel_extras: Tensor, geomBuffer: Tensor, binningBuffer: Tensor,
                                      imgBuffer: Tensor, grad_means2D: Optional[Tensor], grad_conic: Optional[Tensor],
                                      grad_opacity: Optional[Tensor]):
    """Drop-in for ``_C.gaussian_rasterize_extra_backward`` (gaussian_rasterizer_extra.cu:248-277).
    Returns ``(dL_dextra[P,E], dL_dmeans2D[P,3], dL_dconic[P,2,2], dL_dopacity[P,1])``; the three optional inputs are
    accumulated into in place."""
    lib = load_library()
    _require_gpu(extra, 'extras')
    dev = extra.device
    with _on_device(dev):
        extra = _f32c(extra, dev)
        P, E = extra.shape
        f32 = dict(dtype=torch.float32, device=dev)
        gm = grad_means2D if grad_means2D is not None else torch.zeros((P, 3), **f32)
        gc = grad_conic if grad_conic is not None else torch.zeros((P, 2, 2), **f32)
        go = grad_opacity if grad_opacity is not None else torch.zeros((P, 1), **f32)
        ge = torch.empty((P, E), **f32)
        if P == 0:
            return ge, gm, gc, go
        out_opacity = _f32c(out_opacity, dev)
        gpe = _f32c(grad_pixel_extras, dev)
        bufs = _buffers(geomBuffer, binningBuffer, imgBuffer)
        _check(lib.skgs_rasterize_extra_backward(
            C.c_int32(W), C.c_int32(H), C.c_int32(P), C.c_int32(E), C.c_void_p(extra.data_ptr()),
            C.c_void_p(out_opacity.data_ptr()), C.c_void_p(gpe.data_ptr()), C.byref(bufs), C.c_void_p(gm.data_ptr()),
            C.c_void_p(gc.data_ptr()), C.c_void_p(go.data_ptr()), C.c_void_p(ge.data_ptr()), _stream()))
    return ge, gm, gc, go


def gaussian_topk_weights(topk: int, W: int, H: int, P: int, R: int, geomBuffer: Tensor, binningBuffer: Tensor,
                          imgBuffer: Tensor) -> Tuple[Tensor, Tensor]:
    """Drop-in for ``_C.gaussian_topk_weights`` (gaussian_topk.cu:98-121): ``(idx[H,W,k] int32, w[H,W,k])``."""
    lib = load_library()
    _require_gpu(geomBuffer, 'geomBuffer')
    dev = geomBuffer.device
    with _on_device(dev):
        idx = torch.full((H, W, topk), -1, dtype=torch.int32, device=dev)
        w = torch.zeros((H, W, topk), dtype=torch.float32, device=dev)
        if P == 0:
            return idx, w
        bufs = _buffers(geomBuffer, binningBuffer, imgBuffer)
        _check(lib.skgs_topk_weights(C.c_int32(topk), C.c_int32(W), C.c_int32(H), C.c_int32(P), C.byref(bufs),
                                     C.c_void_p(idx.data_ptr()), C.c_void_p(w.data_ptr()), _stream()))
    return idx, w


def render_census(W: int, H: int, geomBuffer: Tensor, binningBuffer: Tensor, imgBuffer: Tensor):
    """Parity-test hook (``skgs_render_census``): the blend forward over the buffers of a finished forward, plus a
    fingerprint of the list entries every pixel blended.  Returns ``(color[3,H,W], opacity[H,W], census[H,W,2] int32)``."""
    lib = load_library()
    _require_gpu(geomBuffer, 'geomBuffer')
    dev = geomBuffer.device
    with _on_device(dev):
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        opacity = torch.empty((H, W), dtype=torch.float32, device=dev)
        census = torch.zeros((H, W, 2), dtype=torch.int32, device=dev)
        bufs = _buffers(geomBuffer, binningBuffer, imgBuffer)
        _check(lib.skgs_render_census(C.c_int32(W), C.c_int32(H), C.byref(bufs), C.c_void_p(color.data_ptr()),
                                      C.c_void_p(opacity.data_ptr()), C.c_void_p(census.data_ptr()), _stream()))
    return color, opacity, census


def mark_visible(positions: Tensor, viewmatrix: Tensor, projmatrix: Tensor, colmap: bool = False) -> Tensor:
    """The reference left ``mark_visible`` commented out (gaussian_rasterizer_imp.cu:75-103) although
    ``GaussianRasterizer.markVisible`` calls it; provided here with the near-plane test of ``in_frustum``."""
    lib = load_library()
    _require_gpu(positions, 'positions')
    dev = positions.device
    with _on_device(dev):
        positions = _f32c(positions, dev)
        viewmatrix = _f32c(viewmatrix, dev)
        P = positions.shape[0]
        out = torch.empty((P,), dtype=torch.uint8, device=dev)
        _check(lib.skgs_mark_visible(C.c_int32(P), C.c_void_p(_ptr(positions)), C.c_void_p(viewmatrix.data_ptr()),
                                     C.c_int32(int(colmap)), C.c_void_p(_ptr(out)), _stream()))
    return out.bool()


# ============================================================================================================= deform
def _deform_inputs(points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit):
    dev = points.device
    ts = [_f32c(t, dev) for t in (points, weights)]
    indices = indices.to(device=dev, dtype=torch.int64).contiguous()
    rest = [_f32c(t, dev) for t in (bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit)]
    a = _DeformInputs()
    a.P, a.K, a.M = ts[0].shape[0], ts[1].shape[1], rest[0].shape[0]
    a.points, a.weights, a.indices = _ptr(ts[0]), _ptr(ts[1]), _ptr(indices)
    (a.bone_T, a.bone_drot, a.bone_dscale, a.xyz, a.log_scale, a.rot, a.opacity_logit) = [_ptr(t) for t in rest]
    return a, ts + [indices] + rest


def lbs_deform_forward(points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit,
                       need_deltas: bool = False):
    """Fused LBS deform + activation epilogue (sk_gs.py:1143-1150,1162,1192-1203).
    Returns ``(means[P,3], scales[P,3], rotations[P,4], opacity[P,1], d_xyz, d_rot, d_scale)`` (deltas None unless
    ``need_deltas``)."""
    lib = load_library()
    _require_gpu(points, 'points')
    dev = points.device
    with _on_device(dev):
        a, keep = _deform_inputs(points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot,
                                 opacity_logit)
        P = a.P
        f32 = dict(dtype=torch.float32, device=dev)
        means, scales = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        rotations, opacity = torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
        d_xyz = torch.empty((P, 3), **f32) if need_deltas else None
        d_rot = torch.empty((P, 4), **f32) if need_deltas else None
        d_scale = torch.empty((P, 3), **f32) if need_deltas else None
        _check(lib.skgs_lbs_deform_forward(C.byref(a), C.c_void_p(_ptr(means)), C.c_void_p(_ptr(scales)),
                                           C.c_void_p(_ptr(rotations)), C.c_void_p(_ptr(opacity)),
                                           C.c_void_p(_ptr(d_xyz)), C.c_void_p(_ptr(d_rot)),
                                           C.c_void_p(_ptr(d_scale)), _stream()))
    return means, scales, rotations, opacity, d_xyz, d_rot, d_scale


_max_fused_bones = None


def fused_lbs_max_bones() -> int:
    """most bones the one-launch KNN + weights (+ skinning) kernels stage in LDS (``skgs_fused_lbs_max_bones``)"""
    global _max_fused_bones
    if _max_fused_bones is None:
        lib = load_library()
        lib.skgs_fused_lbs_max_bones.restype = C.c_int
        _max_fused_bones = int(lib.skgs_fused_lbs_max_bones())
    return _max_fused_bones


def knn_lbs_deform_forward(points, joints, sp_W, K: int, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit):
    """KNN + softmax of the gathered LBS logits + skinning + activation epilogue in ONE launch
    (``skgs_knn_lbs_deform_forward``: what the fused step runs; bit-identical to ``skgs_knn_lbs_weights`` followed by
    ``skgs_lbs_deform_forward``).  The inference path of the `W` weighting (calc_LBS_weight, sk_gs.py:757,767-768, then
    :1143-1150,1162,1192-1203).  Returns ``(means[P,3], scales[P,3], rotations[P,4], opacity[P,1], weights[P,K],
    indices[P,K] int64)``."""
    lib = load_library()
    _require_gpu(points, 'points')
    dev = points.device
    with _on_device(dev):
        ts = [_f32c(t, dev) for t in (points, joints, sp_W, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit)]
        P, M = ts[2].shape
        f32 = dict(dtype=torch.float32, device=dev)
        means, scales = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        rotations, opacity = torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
        weights, indices = torch.empty((P, K), **f32), torch.empty((P, K), dtype=torch.int64, device=dev)
        _check(lib.skgs_knn_lbs_deform_forward(
            C.c_int32(P), C.c_int32(M), C.c_int32(int(K)), *[C.c_void_p(_ptr(t)) for t in ts], C.c_void_p(_ptr(indices)),
            C.c_void_p(_ptr(weights)), C.c_void_p(_ptr(means)), C.c_void_p(_ptr(scales)), C.c_void_p(_ptr(rotations)),
            C.c_void_p(_ptr(opacity)), None, _stream()))
    return means, scales, rotations, opacity, weights, indices


def lbs_deform_backward(points, weights, indices, bone_T, bone_drot, bone_dscale, log_scale, rot, opacity_logit,
                        g_means, g_scales, g_rotations, g_opacity):
    """Returns ``(g_weights[P,K], g_bone_T[M,7], g_bone_drot[M,4], g_bone_dscale[M,3], g_xyz, g_log_scale, g_rot,
    g_opacity_logit)``."""
    lib = load_library()
    _require_gpu(points, 'points')
    dev = points.device
    with _on_device(dev):
        a, keep = _deform_inputs(points, weights, indices, bone_T, bone_drot, bone_dscale, points, log_scale, rot,
                                 opacity_logit)
        P, K, M = a.P, a.K, a.M
        f32 = dict(dtype=torch.float32, device=dev)
        gs = [_f32c(t, dev) for t in (g_means, g_scales, g_rotations, g_opacity)]
        g_weights = torch.empty((P, K), **f32)
        g_bone_T, g_bone_drot = torch.empty((M, 7), **f32), torch.empty((M, 4), **f32)
        g_bone_dscale = torch.empty((M, 3), **f32)
        g_xyz, g_log_scale = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        g_rot, g_op = torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
        lib.skgs_lbs_deform_backward_workspace_bytes.restype = C.c_size_t
        ws = torch.empty((lib.skgs_lbs_deform_backward_workspace_bytes(C.c_int32(P), C.c_int32(M)),), dtype=torch.uint8,
                         device=dev)
        _check(lib.skgs_lbs_deform_backward(
            C.byref(a), C.c_void_p(_ptr(gs[0])), C.c_void_p(_ptr(gs[1])), C.c_void_p(_ptr(gs[2])),
            C.c_void_p(_ptr(gs[3])), C.c_void_p(_ptr(g_weights)), C.c_void_p(g_bone_T.data_ptr()),
            C.c_void_p(g_bone_drot.data_ptr()), C.c_void_p(g_bone_dscale.data_ptr()), C.c_void_p(_ptr(g_xyz)),
            C.c_void_p(_ptr(g_log_scale)), C.c_void_p(_ptr(g_rot)), C.c_void_p(_ptr(g_op)),
            C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _stream()))
    return g_weights, g_bone_T, g_bone_drot, g_bone_dscale, g_xyz, g_log_scale, g_rot, g_op


def knn_bones(points: Tensor, joints: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """K nearest bones (squared L2 ascending) -- the ``pytorch3d.ops.knn_points`` call of sk_gs.py:757.
    Returns ``(dist2[P,K] float32, idx[P,K] int64)``."""
    lib = load_library()
    _require_gpu(points, 'points')
    dev = points.device
    with _on_device(dev):
        points, joints = _f32c(points, dev), _f32c(joints, dev)
        P, dim = points.shape
        M = joints.shape[0]
        dist = torch.empty((P, K), dtype=torch.float32, device=dev)
        idx = torch.empty((P, K), dtype=torch.int64, device=dev)
        _check(lib.skgs_knn_bones(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(dim),
                                  C.c_void_p(_ptr(points)), C.c_void_p(joints.data_ptr()), C.c_void_p(_ptr(dist)),
                                  C.c_void_p(_ptr(idx)), _stream()))
    return dist, idx


def bone_chain_forward(sk_r_raw: Tensor, joints: Tensor, global_T: Optional[Tensor], topo: dict,
                       save_chain: bool = True):
    """``kinematic`` + ``skeleton_warp_SE3`` (sk_gs.py:1069-1107,193-206) in one launch.
    ``topo`` = dict(parents, level_nodes, level_start: int32 device tensors; root, num_levels: int).
    Returns ``(bone_T[M,7], chain_A[M,7] | None)``."""
    lib = load_library()
    _require_gpu(sk_r_raw, 'sk_r_raw')
    dev = sk_r_raw.device
    with _on_device(dev):
        sk_r_raw, joints, global_T = _f32c(sk_r_raw, dev), _f32c(joints, dev), _f32c(global_T, dev)
        M = sk_r_raw.shape[0]
        bone_T = torch.empty((M, 7), dtype=torch.float32, device=dev)
        chain = torch.empty((M, 7), dtype=torch.float32, device=dev) if save_chain else None
        _check(lib.skgs_bone_chain_forward(
            C.c_int32(M), C.c_int32(topo['root']), C.c_void_p(topo['parents'].data_ptr()),
            C.c_void_p(topo['level_nodes'].data_ptr()), C.c_void_p(topo['level_start'].data_ptr()),
            C.c_int32(topo['num_levels']), C.c_void_p(sk_r_raw.data_ptr()), C.c_void_p(joints.data_ptr()),
            C.c_void_p(_ptr(global_T)), C.c_void_p(bone_T.data_ptr()), C.c_void_p(_ptr(chain)), None, _stream()))
    return bone_T, chain


def bone_chain_backward(sk_r_raw: Tensor, joints: Tensor, global_T: Optional[Tensor], topo: dict, chain_A: Tensor,
                        g_bone_T: Tensor, need_joints: bool = False):
    """Returns ``(g_sk_r_raw[M,4], g_joints[M,3] | None, g_global_T[7] | None)``."""
    lib = load_library()
    dev = sk_r_raw.device
    with _on_device(dev):
        sk_r_raw, joints, global_T = _f32c(sk_r_raw, dev), _f32c(joints, dev), _f32c(global_T, dev)
        g_bone_T = _f32c(g_bone_T, dev)
        M = sk_r_raw.shape[0]
        g_raw = torch.empty((M, 4), dtype=torch.float32, device=dev)
        g_j = torch.empty((M, 3), dtype=torch.float32, device=dev) if need_joints else None
        g_g = torch.empty((7,), dtype=torch.float32, device=dev) if global_T is not None else None
        _check(lib.skgs_bone_chain_backward(
            C.c_int32(M), C.c_int32(topo['root']), C.c_void_p(topo['parents'].data_ptr()),
            C.c_void_p(topo['level_nodes'].data_ptr()), C.c_void_p(topo['level_start'].data_ptr()),
            C.c_int32(topo['num_levels']), C.c_void_p(sk_r_raw.data_ptr()), C.c_void_p(joints.data_ptr()),
            C.c_void_p(_ptr(global_T)), C.c_void_p(chain_A.data_ptr()), C.c_void_p(g_bone_T.data_ptr()),
            C.c_void_p(g_raw.data_ptr()), C.c_void_p(_ptr(g_j)), C.c_void_p(_ptr(g_g)), None, _stream()))
    return g_raw, g_j, g_g


def image_loss_forward(pred: Tensor, gt: Tensor, lambda_l1: float, lambda_ssim: float):
    """fused ``lambda_l1 * L1 + lambda_ssim * (1 - SSIM)`` of [C,H,W] images.
    Returns ``(loss3[3] = {total, l1_mean, ssim_mean}, workspace)``."""
    lib = load_library()
    _require_gpu(pred, 'pred')
    dev = pred.device
    with _on_device(dev):
        pred, gt = _f32c(pred, dev), _f32c(gt, dev)
        Cc, H, W = pred.shape
        nbytes = lib.skgs_image_loss_workspace_bytes(C.c_int32(Cc), C.c_int32(H), C.c_int32(W))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        loss3 = torch.empty((3,), dtype=torch.float32, device=dev)
        _check(lib.skgs_image_loss_forward(C.c_int32(Cc), C.c_int32(H), C.c_int32(W), C.c_void_p(pred.data_ptr()),
                                           C.c_void_p(gt.data_ptr()), None, C.c_float(lambda_l1), C.c_float(lambda_ssim),
                                           C.c_void_p(loss3.data_ptr()), C.c_void_p(ws.data_ptr()),
                                           C.c_size_t(nbytes), _stream()))
    return loss3, ws


def image_loss_backward(pred: Tensor, gt: Tensor, lambda_l1: float, lambda_ssim: float, grad_loss: Optional[Tensor],
                        workspace: Tensor) -> Tensor:
    lib = load_library()
    dev = pred.device
    with _on_device(dev):
        pred, gt = _f32c(pred, dev), _f32c(gt, dev)
        Cc, H, W = pred.shape
        out = torch.empty_like(pred)
        gl = _f32c(grad_loss, dev) if grad_loss is not None else None
        _check(lib.skgs_image_loss_backward(C.c_int32(Cc), C.c_int32(H), C.c_int32(W), C.c_void_p(pred.data_ptr()),
                                            C.c_void_p(gt.data_ptr()), None, C.c_float(lambda_l1), C.c_float(lambda_ssim),
                                            C.c_void_p(_ptr(gl)), C.c_void_p(workspace.data_ptr()),
                                            C.c_size_t(workspace.numel()), C.c_void_p(out.data_ptr()), None, _stream()))
    return out


def freq_encode_forward(inputs: Tensor, B: int, D: int, deg: int, C_out: int, outputs: Tensor) -> None:
    """``freq_encode_forward`` of the reference's extension (my_ext/_C/src/nerf/freqencoder.cu:66-85), same positional
    arguments: ``outputs[B, C] = [x | sin(2^f x), cos(2^f x) for f < deg]`` written in place, ``C = D + 2 D deg``.  What
    ``networks/encoders/freq_encoder.py:13,33`` resolves at import time and calls from ``_freq_encoder.forward``."""
    lib = load_library()
    _require_gpu(inputs, 'inputs')
    _require_gpu(outputs, 'outputs')
    _check_freq_args('freq_encode_forward', B, D, deg, C_out, (inputs, B * D, 'inputs'), (outputs, B * C_out, 'outputs'))
    with _on_device(inputs.device):
        _check(lib.skgs_freq_encode_forward(C.c_int32(B), C.c_int32(D), C.c_int32(deg), C.c_void_p(_ptr(inputs)),
                                            C.c_int32(D), C.c_void_p(_ptr(outputs)), C.c_int32(C_out), _stream()))


def freq_encode_backward(grad: Tensor, outputs: Tensor, B: int, D: int, deg: int, C_out: int, grad_inputs: Tensor) -> None:
    """``freq_encode_backward`` (freqencoder.cu:87-105): ``grad_inputs[B, D]`` is WRITTEN (the reference's kernel assigns,
    `:59`) from the cotangent ``grad[B, C]`` and the saved ``outputs[B, C]`` (cos / sin are read back from them)."""
    lib = load_library()
    for t, n in ((grad, 'grad'), (outputs, 'outputs'), (grad_inputs, 'grad_inputs')):
        _require_gpu(t, n)
    _check_freq_args('freq_encode_backward', B, D, deg, C_out, (grad, B * C_out, 'grad'), (outputs, B * C_out, 'outputs'),
                     (grad_inputs, B * D, 'grad_inputs'))
    with _on_device(grad.device):
        _check(lib.skgs_freq_encode_backward(C.c_int32(B), C.c_int32(D), C.c_int32(deg), C.c_void_p(_ptr(grad)),
                                             C.c_void_p(_ptr(outputs)), C.c_int32(C_out), C.c_void_p(_ptr(grad_inputs)),
                                             C.c_int32(0), _stream()))


def simple_knn(points: Tensor) -> Tensor:
    """``simple_knn`` of the reference's extension (my_ext/_C/src/other/knn.cu:192-205; called once by ``create_from_pcd``,
    networks/gaussian_splatting.py:211-213): the mean squared distance of every point to its three nearest other points."""
    lib = load_library()
    _require_gpu(points, 'points')
    if points.ndim != 2 or points.shape[-1] != 3 or points.dtype is not torch.float32:
        raise SkgsError('simple_knn: points must be a float32 tensor of shape (P, 3)')  # the reference's CHECK_TYPE / BCNN_ASSERT
    with _on_device(points.device):
        pts = points.contiguous()
        out = torch.empty((pts.shape[0],), dtype=torch.float32, device=pts.device)
        _check(lib.skgs_simple_knn(C.c_int32(pts.shape[0]), C.c_void_p(_ptr(pts)), C.c_void_p(_ptr(out)), _stream()))
    return out


def _check_freq_args(who, B, D, deg, C_out, *tensors):
    # the reference's CHECK_CONTIGUOUS / CHECK_IS_FLOATING (freqencoder.cu:68-75), plus the sizes its kernels assume
    if C_out != D + 2 * D * deg:
        raise SkgsError(f'{who}: C = {C_out} is not D + 2 D deg = {D + 2 * D * deg}')
    for t, n, name in tensors:
        if t.dtype is not torch.float32 or not t.is_contiguous():
            raise SkgsError(f'{who}: {name} must be a contiguous float32 tensor')
        if t.numel() != n:
            raise SkgsError(f'{who}: {name} has {t.numel()} elements, expected {n}')


#: the names the reference's compiled module ``my_ext._C._C`` defines for THIS path (pybind ``m.def`` names:
#: gaussian_rasterizer_forward.cu:314, gaussian_rasterizer_backwrad.cu:260, gaussian_rasterizer_extra.cu:279-283,
#: gaussian_topk.cu:121, freqencoder.cu:107-110, other/knn.cu:205; ``mark_visible`` is commented out there, gaussian_rasterizer_imp.cu:75-103)
PYBIND_NAMES = ('rasterize_gaussians', 'rasterize_gaussians_backward', 'gaussian_rasterize_extra_forward',
                'gaussian_rasterize_extra_backward', 'gaussian_topk_weights', 'mark_visible', 'freq_encode_forward',
                'freq_encode_backward', 'simple_knn')

_FUNCTIONS = {
    'rasterize_gaussians': rasterize_gaussians,
    'rasterize_gaussians_backward': rasterize_gaussians_backward,
    'gaussian_rasterize_extra_forward': gaussian_rasterize_extra_forward,
    'gaussian_rasterize_extra_backward': gaussian_rasterize_extra_backward,
    'gaussian_topk_weights': gaussian_topk_weights,
    'mark_visible': mark_visible,
    'freq_encode_forward': freq_encode_forward,
    'freq_encode_backward': freq_encode_backward,
    'simple_knn': simple_knn,
    'lbs_deform_forward': lbs_deform_forward,
    'knn_lbs_deform_forward': knn_lbs_deform_forward,
    'lbs_deform_backward': lbs_deform_backward,
    'knn_bones': knn_bones,
}
assert all(n in _FUNCTIONS for n in PYBIND_NAMES)


def get_C_function(name):
    """Same contract as ``my_ext._C.get_C_function`` (my_ext/_C/__init__.py:39-40): ``getattr(_C, name, None)`` for a
    string -- an unknown name yields ``None`` -- and a callable is handed back as it is.  A missing library still raises:
    there is nothing behind the names without it."""
    load_library()
    if not isinstance(name, str):
        return name
    return _FUNCTIONS.get(name)


def have_C_functions(*names) -> bool:
    """``my_ext._C.have_C_functions`` (my_ext/_C/__init__.py:43-47)"""
    return all(n in _FUNCTIONS for n in names)


_pybind_module = None


def pybind_module():
    """A module object shaped like the reference's COMPILED extension ``my_ext._C._C`` (the inner module that
    ``my_ext/_C/__init__.py:14`` imports with ``from . import _C``): its only public attributes are the ops of this path
    under their pybind names, so the reference's own ``hasattr(_C, name)`` / ``getattr(_C, name, None)`` probes
    (my_ext/_C/__init__.py:20,29,40,45) see exactly what a build of the extension with these translation units would
    define and fall back to their Python twins for everything else, as they do for any op a build lacks.  Made once;
    ``sk_gs_amd.install_as_my_ext_C`` registers it as ``sys.modules['my_ext._C._C']``."""
    global _pybind_module
    if _pybind_module is None:
        import types
        m = types.ModuleType('my_ext._C._C', 'sk_gs_amd (MI355X / HIP) ops under the pybind names of my_ext._C._C')
        for n in PYBIND_NAMES:
            setattr(m, n, _FUNCTIONS[n])
        m.__file__ = _LIB_PATH
        _pybind_module = m
    return _pybind_module
