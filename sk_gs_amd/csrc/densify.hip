// densify.hip -- per-iteration densification statistics (scope row (f)-4, first part).
//
// Reference, run after every backward of the densification phase (networks/sk_gs.py:1990-1997,
// networks/gaussian_splatting.py:503-513):
//     mask = radii > 0
//     max_radii2D[mask]        = max(max_radii2D[mask], radii[mask])
//     xyz_gradient_accum[mask] += || viewspace_points.grad[mask, :2] ||_2
//     denom[mask]              += 1
// i.e. three masked index_put / index_select round trips plus a norm (~10 torch kernels).  One streaming kernel here:
// 20 B read + 12 B written per Gaussian.
#include <algorithm>

#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int32_t* __restrict__ radii,
    const float* __restrict__ grad_means2D /*[P,3]*/, float mult, float* __restrict__ xyz_gradient_accum,
    float* __restrict__ denom, float* __restrict__ max_radii2D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r <= 0) return;
  const float gx = grad_means2D[3 * i], gy = grad_means2D[3 * i + 1];
  max_radii2D[i]        = fmaxf(max_radii2D[i], (float) r);
  const float nrm       = sqrtf(gx * gx + gy * gy);
  xyz_gradient_accum[i] = xyz_gradient_accum[i] + (mult == 1.0f ? nrm : mult * nrm);
  denom[i]              = denom[i] + 1.0f;
}

// ---- densification surgery (scope row (f)-4, second part): every per-Gaussian tensor AND its two Adam moments rebuilt
// by ONE launch.  Reference: change_optimizer / prune_points / densification_postfix (networks/gaussian_splatting.py:
// 515-587) index or concatenate each of the 7 parameters, exp_avg and exp_avg_sq one by one (~100 torch launches, several
// host synchronisations per clone / split / prune).  Every such operation is a row gather:
//   dst_t[i, :] = src_t[rows[i], :]               for i < n_keep   (surviving Gaussians keep parameter and moments)
//   dst_t[i, :] = src_t[rows[i], :]  or  0        for i >= n_keep  (new Gaussians copy their parent's parameters; their
//                                                                   moments start from zero, :531-545)
struct RowTensor {
  const float* src;
  float* dst;
  int32_t row_floats;
  int32_t fresh_is_zero;
};
static_assert(sizeof(RowTensor) == 24, "descriptor layout shared with sk_gs_amd/optim.py");

__global__ void __launch_bounds__(256) gather_rows_kernel(const RowTensor* __restrict__ tensors, long long n_out, long long n_keep,
    const int64_t* __restrict__ rows) {
  const RowTensor t = tensors[blockIdx.y];
  const long long total = n_out * t.row_floats;
  const long long stride = (long long) gridDim.x * 256;
  for (long long e = (long long) blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const long long i = e / t.row_floats;
    const int c       = (int) (e - i * t.row_floats);
    t.dst[e] = (i >= n_keep && t.fresh_is_zero) ? 0.f : t.src[rows[i] * t.row_floats + c];
  }
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" size_t skgs_row_tensor_bytes(void) { return sizeof(RowTensor); }

extern "C" int skgs_gather_rows(int32_t n_tensors, const void* tensors, int64_t n_out, int64_t n_keep, const int64_t* rows,
    int32_t max_row_floats, skgs_stream_t stream) {
  SKGS_REQUIRE(n_tensors >= 0 && n_out >= 0 && n_keep >= 0 && n_keep <= n_out && max_row_floats >= 1, "gather_rows: bad sizes");
  if (n_tensors == 0 || n_out == 0) return 0;
  SKGS_REQUIRE(tensors && rows, "gather_rows: NULL argument");
  const long long total = (long long) n_out * max_row_floats;
  const unsigned gx     = (unsigned) std::min<long long>((total + 1023) / 1024, 8192);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, (unsigned) n_tensors), dim3(256), 0, (hipStream_t) stream,
      reinterpret_cast<const RowTensor*>(tensors), (long long) n_out, (long long) n_keep, rows);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int skgs_densify_stats(int32_t P, const int32_t* radii, const float* grad_means2D, float grad_multiplier,
    float* xyz_gradient_accum, float* denom, float* max_radii2D, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0, "densify_stats: P must be >= 0");
  if (P == 0) return 0;
  SKGS_REQUIRE(radii && grad_means2D && xyz_gradient_accum && denom && max_radii2D, "densify_stats: NULL argument");
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t) stream, P, radii, grad_means2D,
      grad_multiplier, xyz_gradient_accum, denom, max_radii2D);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
