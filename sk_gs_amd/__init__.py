"""sk_gs_amd -- MI355X (gfx950) implementation of SK_GS's per-frame hot path: the LBS deform of every Gaussian and the
differentiable 3DGS tile rasterizer, behind the reference's operator surface.

Layout (only what the path needs):
  csrc/                     hand-written HIP kernels + the C ABI (include/skgs.h) -> libskgs_hip.so
  _C.py                     ctypes binding, reference pybind names (get_C_function)
  renderer/gaussian_render  mirror of networks/renderer/gaussian_render.py
  diff_gaussian_rasterization  upstream-compatible front end (variant A of the boundary)
  deform.py / skeleton.py   lbs_deform autograd op, bone chain
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
    """Make ``from my_ext._C import get_C_function`` (networks/renderer/gaussian_render.py:12) resolve to this package's
    binding, so the reference's own renderer module runs unchanged on top of libskgs_hip.so."""
    import sys
    import types
    from sk_gs_amd import _C
    if single_thread:
        single_thread_backward(True)
    pkg = sys.modules.get('my_ext')
    if pkg is None:
        pkg = types.ModuleType('my_ext')
        pkg.__path__ = []
        sys.modules['my_ext'] = pkg
    sys.modules['my_ext._C'] = _C
    pkg._C = _C


def install_as_diff_gaussian_rasterization(single_thread: bool = True):
    """Make ``from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings``
    (networks/renderer/gaussian_render_origin.py:7, networks/gaussian_splatting.py) resolve to this package."""
    from sk_gs_amd import diff_gaussian_rasterization as m
    m.install()
    if single_thread:
        single_thread_backward(True)
