// adam.hip -- multi-tensor Adam step in ONE launch (scope row (f)-2).
//
// Reference: torch.optim.Adam(eps=1e-15, betas=(0.9, 0.999)) over the six Gaussian parameter groups with per-group
// learning rates (exps/default.yaml:122-125, networks/gaussian_splatting.py:443-453), stepped by
// my_ext/framework.py:264-306.  torch's fused path issues one multi_tensor_apply launch per group and state list
// (7 launches, ~300 us per step for config #1 on MI355X = 0.5 TB/s).  The update is a pure stream:
// 16 B read + 12 B written per element.  Here every tensor of every group is walked by one grid with float4 accesses;
// the step counter lives on the device so the launch can sit inside a captured hipGraph.
//
// Math (identical to torch, amsgrad = False, weight_decay = 0, maximize = False):
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <algorithm>
#include <cstdint>

#include "adam_update.h"
#include "skgs_common.h"

namespace skgs {
namespace {

__global__ void __launch_bounds__(ADAM_THREADS) adam_step_kernel(int n_tensors, const AdamTensor* __restrict__ tensors,
    int64_t chunk_begin, int64_t total_chunks, double beta1d, double beta2d, float eps,
    const float* __restrict__ step_count) {
  const AdamCoef k = adam_coefficients(beta1d, beta2d, eps, step_count);
  const int lane = threadIdx.x & 63;
  const int64_t first0 = lane < n_tensors ? tensors[lane].chunk0 : INT64_MAX;
  for (int64_t chunk = chunk_begin + blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    const AdamTensor T = tensors[adam_owner(tensors, n_tensors, first0, lane, chunk)];
    adam_update_chunk(T, (chunk - T.chunk0) * ADAM_CHUNK, threadIdx.x, k);
  }
}

// (A last-workgroup-out ticket inside adam_step_kernel was tried instead of this launch: 4096 same-address atomics
// next to the counter every workgroup reads cost 70 us.)
// ... and, in the same launch, clears `zero_after` (gradient storage that must read zero when the next backward starts:
// the per-frame tables of which a step writes one row -- instead of a fill launch at the start of every step)
__global__ void __launch_bounds__(256) adam_bump_kernel(float* step_count, float* __restrict__ zero_after, int64_t zero_n) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i == 0) step_count[0] += 1.0f;
  if (i < zero_n) zero_after[i] = 0.f;
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_adam_tensor_bytes(void) { return sizeof(AdamTensor); }
int64_t skgs_adam_chunk_elems(void) { return ADAM_CHUNK; }

/* tensors: DEVICE array of n_tensors descriptors {param, grad, exp_avg, exp_avg_sq, n, chunk0, lr} (56 B each, every
 * pointer 16-B aligned; chunk0 = running sum of ceil(n / skgs_adam_chunk_elems())). step_count: device float, the
 * number of steps taken so far; incremented by the call. */
int skgs_adam_step(int32_t n_tensors, const void* tensors, int64_t total_chunks, double beta1, double beta2, double eps,
    float* step_count, float* zero_after, int64_t zero_n, skgs_stream_t stream) {
  if (n_tensors == 0 || total_chunks == 0) return 0;
  return skgs_adam_step_range(n_tensors, tensors, 0, total_chunks, beta1, beta2, eps, step_count, 1, zero_after, zero_n,
      stream);
}

/* One step taken in pieces: the chunks [chunk_begin, chunk_end) of the table (whole tensors: the chunk0 of a tensor and of
 * the one after it) are updated with the bias correction of step *step_count + 1; the counter moves (and zero_after is
 * cleared) only where `advance` is set -- in the LAST piece, ordered after all the others.  Pieces of one step may run on
 * different streams, beside the backward kernels that do not touch their tensors.  chunk_begin == chunk_end with
 * advance = 1 only moves the counter. */
int skgs_adam_step_range(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1,
    double beta2, double eps, float* step_count, int32_t advance, float* zero_after, int64_t zero_n, skgs_stream_t stream) {
  SKGS_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors) && step_count, "adam_step: NULL argument");
  SKGS_REQUIRE(chunk_begin >= 0 && chunk_end >= chunk_begin, "adam_step: bad chunk range");
  hipStream_t s = (hipStream_t) stream;
  if (n_tensors > 0 && chunk_end > chunk_begin) {
    const int grid = (int) std::min<int64_t>(chunk_end - chunk_begin, 256 * 16);
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(ADAM_THREADS), 0, s, n_tensors,
        reinterpret_cast<const AdamTensor*>(tensors), chunk_begin, chunk_end, beta1, beta2, (float) eps, step_count);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  if (advance) {
    const int64_t zn = zero_after ? std::max<int64_t>(zero_n, 0) : 0;
    hipLaunchKernelGGL(adam_bump_kernel, dim3((unsigned) std::max<int64_t>(1, (zn + 255) / 256)), dim3(256), 0, s,
        step_count, zero_after, zn);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // extern "C"
