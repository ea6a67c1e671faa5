// weight_reg.hip -- the two per-Gaussian regularisers the shipped stage-sp configuration runs on the LBS weights [P,K] in EVERY iteration
// (exps/default.yaml:85-86 `sparse: 0.1`, `smooth: 0.1`; networks/sk_gs.py:1572-1574), each as ONE launch that produces the value AND the
// gradient:
//
//   loss_weight_sparsity (sk_gs.py:1339-1340):  -mean( w log(w + eps) + (1 - w) log(1 - w + eps) )                       over P K elements
//   loss_weight_smooth   (sk_gs.py:1357-1359):   mean | w[i, k] - w[nbr[i, g], k] |   over i < P, g < G (gs_knn_num + 1 = 21), k < K
//
// As torch writes them the second is a [P, G, K] gather (42 MB at P = 1e5) whose backward is torch's sort-based index_put: ~1 ms of the
// iteration, twice what the whole fused step takes.  Here: a lane per (Gaussian, neighbour slot) pair; the value's partial sums go through
// a fixed-order tree per workgroup into `partials` (the caller adds them: bitwise reproducible), the gradient of the smooth term into a
// zero-filled [P,K] buffer by float atomics (its order-dependent bits are those of every other atomic sum on this path).
// The gradients are written UNSCALED by the incoming cotangent (d value / d w): the autograd node multiplies.
#include <algorithm>

#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int WR_THREADS = 256;

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  // fixed order: lanes of a wave by xor shuffles, then the four waves in index order
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) s = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(WR_THREADS) weight_sparsity_kernel(long long n, const float* __restrict__ w, float eps, float inv_n,
    float* __restrict__ grad, float* __restrict__ partials) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (long long i = (long long) blockIdx.x * WR_THREADS + threadIdx.x; i < n; i += (long long) gridDim.x * WR_THREADS) {
    const float x = w[i];
    const float a = x + eps, b = (1.f - x) + eps;
    const float la = logf(a), lb = logf(b);
    acc += x * la + (1.f - x) * lb;
    // d/dx [ x log(x + eps) + (1 - x) log(1 - x + eps) ] = log a + x / a - log b - (1 - x) / b ;  the loss is MINUS the mean
    grad[i] = -inv_n * ((la + x / a) - (lb + (1.f - x) / b));
  }
  const float s = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = -inv_n * s;
}

// one lane per (i, g): K <= 16 weights of row i against row nbr[i, g]
__global__ void __launch_bounds__(WR_THREADS) weight_smooth_kernel(int P, int K, int G, const float* __restrict__ w,
    const int64_t* __restrict__ nbr, float inv_n, float* __restrict__ grad /* zero-filled */, float* __restrict__ partials) {
  __shared__ float s_red[4];
  float acc = 0.f;
  const long long pairs = (long long) P * G;
  for (long long e = (long long) blockIdx.x * WR_THREADS + threadIdx.x; e < pairs; e += (long long) gridDim.x * WR_THREADS) {
    const int i = (int) (e / G);
    long long j = nbr[e];
    if (j < 0) j += P;                       // torch's index semantics
    if (j < 0 || j >= P || j == i) continue; // (a wild index addresses nothing; a Gaussian against itself contributes 0 and no gradient)
    for (int k = 0; k < K; ++k) {
      const float d = w[(size_t) i * K + k] - w[(size_t) j * K + k];
      acc += fabsf(d);
      const float sg = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
      if (sg != 0.f) {
        atomicAdd(grad + (size_t) i * K + k, sg);
        atomicAdd(grad + (size_t) j * K + k, -sg);
      }
    }
  }
  const float s = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = inv_n * s;
}

// No atomics: a lane per Gaussian i sums sign(w_i - w_j) over its OWN G neighbours (and their |.| for the value), then MINUS sign(w_i' - w_i)
// over the Gaussians i' that list i (inverse lists, CSR: built once per neighbour table by the caller) -- the [P,K] weight table is 2 MB and
// stays in L2.  (The atomic form above: 21 M float atomics at P = 1e5, G = 21, K = 5 -- 1.2 ms, slower than torch's gather + index_put.)
template <int KMAX>
__global__ void __launch_bounds__(WR_THREADS) weight_smooth_lists_kernel(int P, int K, int G, const float* __restrict__ w,
    const int64_t* __restrict__ nbr, const int32_t* __restrict__ inv_off, const int32_t* __restrict__ inv_src, float inv_n,
    float* __restrict__ grad, float* __restrict__ partials) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int i = blockIdx.x * WR_THREADS + threadIdx.x; i < P; i += gridDim.x * WR_THREADS) {
    float wi[KMAX], gi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wi[k] = k < K ? w[(size_t) i * K + k] : 0.f, gi[k] = 0.f;
    for (int g = 0; g < G; ++g) {
      long long j = nbr[(size_t) i * G + g];
      if (j < 0) j += P;
      if (j < 0 || j >= P || j == i) continue;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          const float d = wi[k] - w[(size_t) j * K + k];
          acc += fabsf(d);
          gi[k] += d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
    }
    for (int e = inv_off[i]; e < inv_off[i + 1]; ++e) {
      const int s = inv_src[e];
      if (s == i) continue;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          const float d = w[(size_t) s * K + k] - wi[k];
          gi[k] -= d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) grad[(size_t) i * K + k] = inv_n * gi[k];
  }
  const float s = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = inv_n * s;
}

inline int wr_grid(long long n) { return (int) std::max<long long>(1, std::min<long long>((n + WR_THREADS - 1) / WR_THREADS, 2048)); }

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

int32_t skgs_weight_reg_partials(void) { return 2048; }

int skgs_weight_sparsity(int64_t n, const float* weights, float eps, float* grad, float* partials, skgs_stream_t stream) {
  SKGS_REQUIRE(n >= 0 && (n == 0 || (weights && grad)) && partials, "weight_sparsity: bad argument");
  hipStream_t s = (hipStream_t) stream;
  const int g   = wr_grid(n);
  if (fill_u32(partials, 0u, 2048, s)) return 1;   // (a kernel, not a memset node: the launches stay capturable as plain kernel nodes)
  if (n > 0) hipLaunchKernelGGL(weight_sparsity_kernel, dim3(g), dim3(WR_THREADS), 0, s, (long long) n, weights, eps, 1.0f / (float) n, grad, partials);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_weight_smooth(int32_t P, int32_t K, int32_t G, const float* weights, const int64_t* neighbours, const int32_t* inverse_offsets,
    const int32_t* inverse_sources, float* grad, float* partials, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && K >= 1 && K <= 16 && G >= 1 && (P == 0 || (weights && neighbours && grad)) && partials, "weight_smooth: bad argument");
  SKGS_REQUIRE((inverse_offsets == nullptr) == (inverse_sources == nullptr), "weight_smooth: inverse lists come as a pair");
  hipStream_t s = (hipStream_t) stream;
  if (fill_u32(partials, 0u, 2048, s)) return 1;
  if (P > 0) {
    const float inv_n = 1.0f / ((float) P * (float) G * (float) K);
    if (inverse_offsets) {
      if (K <= 8)
        hipLaunchKernelGGL(weight_smooth_lists_kernel<8>, dim3(wr_grid(P)), dim3(WR_THREADS), 0, s, P, K, G, weights, neighbours, inverse_offsets,
            inverse_sources, inv_n, grad, partials);
      else
        hipLaunchKernelGGL(weight_smooth_lists_kernel<16>, dim3(wr_grid(P)), dim3(WR_THREADS), 0, s, P, K, G, weights, neighbours, inverse_offsets,
            inverse_sources, inv_n, grad, partials);
    } else {
      if (fill_u32(grad, 0u, (size_t) P * K, s)) return 1;
      hipLaunchKernelGGL(weight_smooth_kernel, dim3(wr_grid((long long) P * G)), dim3(WR_THREADS), 0, s, P, K, G, weights, neighbours, inv_n, grad,
          partials);
    }
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
