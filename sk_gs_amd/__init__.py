"""sk_gs_amd -- MI355X (gfx950) implementation of SK_GS's per-frame hot path: the LBS deform of every Gaussian and the
differentiable 3DGS tile rasterizer, behind the reference's operator surface.

Layout (only what the path needs):
  csrc/                     hand-written HIP kernels + the C ABI (include/skgs.h) -> libskgs_hip.so
  _C.py                     ctypes binding, reference pybind names (get_C_function)
  renderer/gaussian_render  mirror of networks/renderer/gaussian_render.py
  diff_gaussian_rasterization  upstream-compatible front end (variant A of the boundary)
  deform.py / skeleton.py   lbs_deform autograd op, bone chain
  lietorch.py / pytorch3d_ops.py  stand-ins for the two CUDA-only packages networks/sk_gs.py imports (SE3 / SO3, knn_points): the
                            reference's deform runs UNMODIFIED on them, its skinning expression and search as launches of the library
  scene.py / model.py       synthetic scenes and the minimal skinned-Gaussian module used by tests and bench
  view_parallel.py          one-process-per-GPU view-parallel gradient all-reduce (RCCL)
"""
__version__ = '0.1.0'


def single_thread_backward(on: bool = True):
    """Run autograd's backward on the thread that calls ``.backward()`` instead of torch's per-device worker thread
    (``torch.autograd.set_multithreading_enabled``).  One process drives one GPU here, so the worker thread buys nothing, and
    its hand-off is the largest single host cost of an operator-path training step on the MI355X boxes' hosts: 0.92 -> 0.57 ms
    of host time per step (``tools/prof_autograd_step.py``).  Same arithmetic, same order of the nodes.  The two
    ``install_as_*`` hooks switch it on unless told otherwise; ``single_thread_backward(False)`` restores torch's default."""
    import torch
    torch.autograd.set_multithreading_enabled(not on)


def install_as_my_ext_C(single_thread: bool = True):
    """Put this package's ops behind the reference's extension lookup, so that the reference's own modules --
    ``networks/renderer/gaussian_render.py`` (``get_C_function('rasterize_gaussians')`` ...), ``networks/encoders/
    freq_encoder.py:13-14`` (``freq_encode_forward/backward``) -- run unmodified on top of libskgs_hip.so.

    In the reference ``my_ext._C`` is a Python PACKAGE (my_ext/_C/__init__.py) that defines ``get_C_function``,
    ``try_use_C_extension``, ``have_C_functions``, ``check_C_runtime`` and ``get_python_function`` over the compiled
    module it imports with ``from . import _C`` (:14) -- the INNER module ``my_ext._C._C``.  That inner module is the one
    thing replaced: ``sys.modules['my_ext._C._C'] = sk_gs_amd._C.pybind_module()``.  The reference's package, its helper
    functions and every Python twin stay the reference's own; ops this library does not define (``xfm_fwd``,
    ``quaternion_to_R_forward``, ...) are reported missing by the reference's own probes and fall back to their Python
    implementations exactly as with a partial build of its extension.

    Call it BEFORE the first ``import my_ext`` / ``import networks``: the reference resolves ops at import time
    (freq_encoder.py:13-14, cdist_top.py:42-44) and, once its ``my_ext._C`` has been imported without a compiled module,
    has already taken the "Please Compile" branch (my_ext/_C/__init__.py:104-130) -- that is refused here with an error.

    Where the reference's ``my_ext`` cannot be found at all (a machine that only has this repository: the GPU box's
    tests, a stand-alone copy of the renderer module), a minimal stand-in for the two outer packages is planted so that
    ``from my_ext._C import get_C_function`` still resolves (``sk_gs_amd._C`` offers the same helper names)."""
    import importlib.util
    import sys
    import types
    from sk_gs_amd import _C
    inner = _C.pybind_module()
    outer = sys.modules.get('my_ext._C')
    if outer is not None and getattr(outer, '_sk_gs_amd_stand_in', False):
        pass  # our own stand-in from an earlier call
    elif outer is not None and getattr(outer, '_C', None) is not inner:
        raise RuntimeError(
            "sk_gs_amd.install_as_my_ext_C(): the reference's my_ext._C is already imported without these ops (its modules "
            "resolved their C functions at import time) -- call install_as_my_ext_C() before importing my_ext / networks")
    if single_thread:
        single_thread_backward(True)
    sys.modules['my_ext._C._C'] = inner
    if outer is not None:
        return
    try:
        found = 'my_ext' in sys.modules or importlib.util.find_spec('my_ext') is not None
    except (ImportError, ValueError):
        found = False
    if found:
        return  # the reference is importable: its own my_ext/_C/__init__.py will pick the inner module up
    pkg = types.ModuleType('my_ext')
    pkg.__path__ = []
    pkg._sk_gs_amd_stand_in = True
    mid = types.ModuleType('my_ext._C')
    mid.__path__ = []
    mid._sk_gs_amd_stand_in = True
    mid._C = inner
    # the helper names of my_ext/_C/__init__.py:9-48 over the inner module, with the reference's semantics
    mid.get_C_function = lambda func: getattr(inner, func, None) if isinstance(func, str) else func
    mid.have_C_functions = lambda *names: all(hasattr(inner, n) for n in names)
    pkg._C = mid
    pkg.get_C_function, pkg.have_C_functions = mid.get_C_function, mid.have_C_functions
    sys.modules['my_ext'] = pkg
    sys.modules['my_ext._C'] = mid


def uninstall_my_ext_C():
    """Undo ``install_as_my_ext_C`` for modules not imported yet (tests): drops the inner module and any stand-in."""
    import sys
    sys.modules.pop('my_ext._C._C', None)
    for name in ('my_ext._C', 'my_ext'):
        m = sys.modules.get(name)
        if m is not None and getattr(m, '_sk_gs_amd_stand_in', False):
            del sys.modules[name]


def install_as_lietorch():
    """``from lietorch import SE3, SO3`` (networks/sk_gs.py:12, networks/gaussian_splatting.py:27, networks/GS_utils.py:7) resolves to
    ``sk_gs_amd.lietorch``: the reference's deform -- ``skeleton_warp_SE3``, ``kinematic``, ``warp``, ``sk_stage``, ``sp_stage``, its
    Lie-group losses -- runs unmodified, pure torch on any device, with the per-Gaussian skinning expression
    ``(sk_T[indices].act(points[:, None]) * weights[..., None]).sum(dim=1)`` as one launch of libskgs_hip.so on a HIP device.  Refuses
    to shadow a real lietorch that is already imported."""
    import sys
    from sk_gs_amd import lietorch as m
    have = sys.modules.get('lietorch')
    if have is not None and have is not m:
        raise RuntimeError('sk_gs_amd.install_as_lietorch(): a different `lietorch` module is already imported')
    sys.modules['lietorch'] = m
    return m


def install_as_pytorch3d():
    """``from pytorch3d.ops import knn_points`` (networks/sk_gs.py:11; ``import pytorch3d.ops`` / ``ball_query`` in
    networks/losses/SC_GS_arap_loss.py:5-7) resolves to ``sk_gs_amd.pytorch3d_ops``: ``knn_points`` of calc_LBS_weight
    (sk_gs.py:757) is one launch of libskgs_hip.so on a HIP device.  Refuses to shadow a real pytorch3d that is already imported."""
    import sys
    import types
    from sk_gs_amd import pytorch3d_ops as ops
    have = sys.modules.get('pytorch3d')
    if have is not None and not getattr(have, '_sk_gs_amd_stand_in', False):
        raise RuntimeError('sk_gs_amd.install_as_pytorch3d(): a different `pytorch3d` package is already imported')
    if have is None:
        pkg = types.ModuleType('pytorch3d')
        pkg.__path__ = []
        pkg._sk_gs_amd_stand_in = True
        pkg.__version__ = '0.0.0+sk_gs_amd'
        pkg.ops = ops
        sys.modules['pytorch3d'] = pkg
    sys.modules['pytorch3d.ops'] = ops
    return ops


def install_reference_hooks(single_thread: bool = True, accelerate: bool = False):
    """Everything an unmodified checkout of the reference needs from this package on an MI355X machine, in one call made BEFORE
    ``import my_ext`` / ``import networks`` / ``import train``: the compiled ops behind ``my_ext._C`` (``install_as_my_ext_C``), the
    shipped configs' rasterizer package (``install_as_diff_gaussian_rasterization``), and the two CUDA-only third-party packages of
    the deform (``install_as_lietorch``, ``install_as_pytorch3d``).  ``accelerate=True``: ``accelerate_reference()``'s fast paths are
    applied as the reference's modules arrive (no second call).  INTEGRATION.md section 1."""
    install_as_my_ext_C(single_thread=single_thread)
    install_as_diff_gaussian_rasterization(single_thread=single_thread)
    install_as_lietorch()
    install_as_pytorch3d()
    if accelerate:
        # ... and the fast paths of accelerate_reference() applied BY THEMSELVES as the reference's modules finish importing (a post-import
        # hook): the two lines `import sk_gs_amd; sk_gs_amd.install_reference_hooks(accelerate=True)` in front of the reference's own
        # imports are then everything
        from sk_gs_amd import reference_accel
        reference_accel.install_post_import_patcher()
    return ['my_ext._C._C', 'diff_gaussian_rasterization', 'lietorch', 'pytorch3d.ops']


def accelerate_reference(ssim: bool = True, kinematic_chain: bool = True, networks: bool = True, lbs_weights: bool = True,
                         adam: bool = True, swizzle: bool = True, fused_render: bool = True) -> list:
    """AFTER the reference has been imported (``install_reference_hooks()`` before it): give methods of its classes a fast path --
    ``SSIM_Loss.forward`` (3.5 ms of depth-wise convolutions per image -> the fused loss kernels),
    ``SkeletonGaussianSplatting.kinematic`` (~60 Lie-group launches -> one bone-chain launch per direction),
    ``SkeletonGaussianSplatting.calc_LBS_weight`` (search + weighting -> one launch per direction) and the two deform
    networks' ``forward`` (``SimpleDeformationNetwork`` -> the one-launch 20-row network kernels, ``DeformNetwork`` -> the MFMA
    row-block kernels; both on the reference modules' OWN parameter objects) -- and ``torch.optim.Adam.step`` (one launch over the
    optimizer's own state tensors instead of ~80; ``adam=False`` leaves torch alone) and the rasterizer adapter's quaternion swizzle (two
    slices instead of an index list: call this BEFORE the model is built).  Same arguments, same returned objects, the reference's
    own method for every call outside the fast path's conditions; ``sk_gs_amd.reference_accel``.

    ``fused_render`` (round 6): ``SkeletonGaussianSplatting.render`` + ``ImageLoss.forward`` + ``SSIM_Loss.forward`` put the package's
    WHOLE fused per-view step (11 launches forward + backward, the trainer's own) behind the reference's iteration for stage ``sk`` on
    the model's own parameters -- ``sk_gs_amd.reference_fused``; calls outside its conditions run the reference's ``render`` and, inside
    it, the per-method fast paths above."""
    from sk_gs_amd import reference_accel
    return reference_accel.accelerate_reference(ssim=ssim, kinematic_chain=kinematic_chain, networks=networks, lbs_weights=lbs_weights, adam=adam,
                                                swizzle=swizzle, fused_render=fused_render)


def install_as_diff_gaussian_rasterization(single_thread: bool = True):
    """Make ``from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings``
    (networks/renderer/gaussian_render_origin.py:7, networks/gaussian_splatting.py) resolve to this package."""
    from sk_gs_amd import diff_gaussian_rasterization as m
    m.install()
    if single_thread:
        single_thread_backward(True)
