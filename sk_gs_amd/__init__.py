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
