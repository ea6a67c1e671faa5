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

}  // namespace
}  // namespace skgs

using namespace skgs;

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
