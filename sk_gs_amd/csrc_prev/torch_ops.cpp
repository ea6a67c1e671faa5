// torch_ops.cpp -- the operator path's two hot entry points as C++ functions over torch tensors.
//
// The reference's plugin boundary for the rasterizer is a pybind module of C++ functions taking tensors
// (my_ext/_C: `rasterize_gaussians`, `rasterize_gaussians_backward`, resolved through get_C_function,
// my_ext/_C/__init__.py:39-48; signatures gaussian_rasterizer_forward.cu:260-317, gaussian_rasterizer_backwrad.cu:200-261).
// sk_gs_amd/_C.py implements the same two functions in Python over ctypes; at config #1 that marshalling (two structs of
// ~40 fields, ten torch.empty calls, dtype / layout checks) costs more host time per render than the four + three kernel
// launches it leads to.  This file is the same marshalling in C++: tensors in, one C-ABI call of libskgs_hip.so
// (include/skgs.h), tensors out.  No HIP code and no HIP headers: the stream arrives as an integer handle.
//
// Only the sync-free forward (config.sync_num_rendered = False) and the backward live here; the reference-exact forward
// with its host read-back of num_rendered stays in Python (it waits for the device anyway).  sk_gs_amd/_C.py uses this
// module when it is built and its own ctypes path otherwise -- both end in the same C-ABI calls.
#include <torch/extension.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/skgs.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void check(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("skgs: ") + skgs_last_error());
}

// float32, contiguous, on `dev` (the common case converts nothing); empty tensors pass through (their pointer reads NULL)
Tensor prep(const Tensor& t, const at::Device& dev) {
  if (t.numel() == 0) return t;
  if (t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == dev) return t;
  return t.to(dev, at::kFloat).contiguous();
}
const float* fptr(const Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
const float* fptr(const OptTensor& t) { return t.has_value() && t->numel() ? t->data_ptr<float>() : nullptr; }

struct Inputs {
  skgs_raster_inputs a{};
  std::vector<Tensor> keep;  // converted tensors stay alive until the launches are issued
  int P = 0, M = 0, E = 0;
};

void fill_inputs(Inputs& in, int64_t H, int64_t W, double tanfovx, double tanfovy, int64_t degree, double scale_modifier,
    bool prefiltered, bool debug, bool colmap, const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos,
    const Tensor& means3D, const Tensor& opacity, const Tensor& sh, const Tensor& scales, const Tensor& rotations,
    const OptTensor& extras, const Tensor& colors, const Tensor& cov3D) {
  TORCH_CHECK(means3D.is_cuda(), "means3D must live on a HIP device (got ", means3D.device(), "); sk_gs_amd has no CPU path");
  TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
  const at::Device dev = means3D.device();
  auto p = [&](const Tensor& t) -> const float* {
    in.keep.push_back(prep(t, dev));
    return fptr(in.keep.back());
  };
  skgs_raster_inputs& a = in.a;
  a.means3D = p(means3D);
  in.P = (int) means3D.size(0);
  a.sh = p(sh);
  in.M = sh.numel() > 0 ? (int) sh.size(1) : 0;
  if (extras.has_value()) {
    a.extras = p(*extras);
    in.E = extras->numel() > 0 ? (int) extras->size(-1) : 0;
  }
  a.P = in.P, a.sh_degree = (int32_t) degree, a.sh_coeffs = in.M, a.E = in.E;
  a.image_height = (int32_t) H, a.image_width = (int32_t) W;
  a.tanfovx = (float) tanfovx, a.tanfovy = (float) tanfovy, a.scale_modifier = (float) scale_modifier;
  a.prefiltered = prefiltered, a.debug = debug, a.colmap = colmap;
  a.viewmatrix = p(viewmatrix), a.projmatrix = p(projmatrix), a.campos = p(campos);
  a.opacity = p(opacity), a.scales = p(scales), a.rotations = p(rotations);
  a.colors_precomp = p(colors), a.cov3D_precomp = p(cov3D);
}

skgs_raster_buffers buffers(const Tensor& geom, const Tensor& binning, const Tensor& img) {
  skgs_raster_buffers b{};
  b.geom = geom.data_ptr(), b.geom_bytes = (size_t) geom.numel();
  b.binning = binning.numel() ? binning.data_ptr() : nullptr, b.binning_bytes = (size_t) binning.numel();
  b.img = img.data_ptr(), b.img_bytes = (size_t) img.numel();
  return b;
}

// sync-free forward: (color[3,H,W], opacity[H,W], radii[P], geomBuffer, binningBuffer, imgBuffer, out_extras | None)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, OptTensor> rasterize_forward(int64_t H, int64_t W, double tanfovx,
    double tanfovy, int64_t degree, double scale_modifier, bool prefiltered, bool debug, bool colmap, const Tensor& viewmatrix,
    const Tensor& projmatrix, const Tensor& campos, const Tensor& means3D, const Tensor& opacity, const Tensor& sh,
    const Tensor& scales, const Tensor& rotations, const OptTensor& extras, const Tensor& colors, const Tensor& cov3D,
    int64_t geom_bytes, int64_t img_bytes, int64_t binning_bytes, int64_t tile_bucket, int64_t stream) {
  Inputs in;
  fill_inputs(in, H, W, tanfovx, tanfovy, degree, scale_modifier, prefiltered, debug, colmap, viewmatrix, projmatrix, campos,
      means3D, opacity, sh, scales, rotations, extras, colors, cov3D);
  TORCH_CHECK(in.P > 0, "rasterize_forward: the empty scene is handled by the caller");
  const auto f32 = means3D.options().dtype(at::kFloat);
  const auto u8  = means3D.options().dtype(at::kByte);
  Tensor color = at::empty({3, H, W}, f32), opac = at::empty({H, W}, f32);
  Tensor radii = at::empty({in.P}, means3D.options().dtype(at::kInt));
  OptTensor out_extras;
  if (extras.has_value() && in.E > 0) out_extras = at::empty({in.E, H, W}, f32);
  Tensor geom = at::empty({geom_bytes}, u8), img = at::empty({img_bytes}, u8), binning = at::empty({binning_bytes}, u8);
  in.a.tile_bucket_capacity = (int32_t) tile_bucket;
  const skgs_raster_buffers b = buffers(geom, binning, img);
  check(skgs_rasterize_forward(&in.a, &b, radii.data_ptr<int32_t>(), color.data_ptr<float>(), opac.data_ptr<float>(),
      out_extras.has_value() ? out_extras->data_ptr<float>() : nullptr, nullptr, (skgs_stream_t) stream));
  return {color, opac, radii, geom, binning, img, out_extras};
}

// backward: (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dextras | None)
// `workspace`: P * 64 bytes, all zero on entry, handed back all zero (skgs_raster_grads.workspace_is_zero)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, OptTensor> rasterize_backward(double scale_modifier,
    double tanfovx, double tanfovy, int64_t degree, bool debug, bool colmap, const Tensor& viewmatrix, const Tensor& projmatrix,
    const Tensor& campos, const Tensor& means3D, const Tensor& colors, const OptTensor& extras, const Tensor& scales,
    const Tensor& rotations, const Tensor& cov3D, const Tensor& sh, const Tensor& radii, const Tensor& out_opacity,
    const Tensor& dL_dout_color, const Tensor& dL_dout_opacity, const OptTensor& dL_dout_extra, const OptTensor& grad_means2D,
    const OptTensor& grad_conic, const OptTensor& grad_opacity, const Tensor& geom, const Tensor& binning, const Tensor& img,
    const Tensor& workspace, int64_t stream) {
  const int64_t H = dL_dout_color.size(1), W = dL_dout_color.size(2);
  Inputs in;
  // opacity is not an input of the backward (the blend kernels read it from the saved records): any non-null pointer
  fill_inputs(in, H, W, tanfovx, tanfovy, degree, scale_modifier, false, debug, colmap, viewmatrix, projmatrix, campos, means3D,
      means3D, sh, scales, rotations, extras, colors, cov3D);
  const at::Device dev = means3D.device();
  const auto f32 = means3D.options().dtype(at::kFloat);
  const int64_t P = in.P, M = in.M, E = in.E;
  const bool use_extra = extras.has_value() && dL_dout_extra.has_value() && E > 0;
  Tensor dL_dmeans2D = at::empty({P, 3}, f32), dL_dcolors = at::empty({P, 3}, f32), dL_dopacity = at::empty({P, 1}, f32);
  Tensor dL_dmeans3D = at::empty({P, 3}, f32), dL_dcov3D = at::empty({P, 6}, f32), dL_dsh = at::empty({P, M, 3}, f32);
  Tensor dL_dscales = at::empty({P, 3}, f32), dL_drot = at::empty({P, 4}, f32);
  OptTensor dL_dextras;
  if (use_extra) dL_dextras = at::empty({P, E}, f32);
  if (P == 0) return {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot, dL_dextras};
  std::vector<Tensor> keep;
  auto p = [&](const OptTensor& t) -> const float* {
    if (!t.has_value()) return nullptr;
    keep.push_back(prep(*t, dev));
    return fptr(keep.back());
  };
  skgs_raster_grads g{};
  g.dL_dout_color   = p(dL_dout_color);
  g.dL_dout_opacity = p(dL_dout_opacity);
  g.dL_dout_extra   = use_extra ? p(dL_dout_extra) : nullptr;
  g.grad_means2D_in = p(grad_means2D), g.grad_conic_in = p(grad_conic), g.grad_opacity_in = p(grad_opacity);
  const float* opac = p(out_opacity);
  const Tensor radii_c = radii.contiguous();
  g.dL_dmeans2D = dL_dmeans2D.data_ptr<float>(), g.dL_dcolors = dL_dcolors.data_ptr<float>();
  g.dL_dopacity = dL_dopacity.data_ptr<float>(), g.dL_dmeans3D = dL_dmeans3D.data_ptr<float>();
  g.dL_dcov3D = dL_dcov3D.data_ptr<float>(), g.dL_dsh = M ? dL_dsh.data_ptr<float>() : nullptr;
  g.dL_dscales = dL_dscales.data_ptr<float>(), g.dL_drotations = dL_drot.data_ptr<float>();
  g.dL_dextras = use_extra ? dL_dextras->data_ptr<float>() : nullptr;
  g.workspace = reinterpret_cast<float*>(workspace.data_ptr()), g.workspace_bytes = (size_t) workspace.numel();
  g.workspace_is_zero = 1;
  if (!use_extra) in.a.extras = nullptr, in.a.E = 0;
  const skgs_raster_buffers b = buffers(geom, binning, img);
  check(skgs_rasterize_backward(&in.a, &b, radii_c.data_ptr<int32_t>(), opac, &g, (skgs_stream_t) stream));
  return {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot, dL_dextras};
}

}  // namespace

PYBIND11_MODULE(_skgs_torch, m) {
  m.doc() = "tensor-level entry points of the rasterizer over libskgs_hip.so (see sk_gs_amd/_C.py)";
  m.def("rasterize_forward", &rasterize_forward);
  m.def("rasterize_backward", &rasterize_backward);
}
