// adam_update.h -- the Adam update of one chunk of one tensor, shared by the optimizer's own launch (adam.hip) and the
// side job of the deform network's backward launch (mlp_fused.hip: the Gaussian rows' update runs on the CUs that launch
// leaves idle).  Math: see adam.hip.
#pragma once
#include <cstdint>

#include "skgs_common.h"

namespace skgs {

struct AdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;        // elements
  int64_t chunk0;   // first chunk index of this tensor in the flattened chunk space
  float lr;
  float pad;
};
static_assert(sizeof(AdamTensor) == 56, "layout shared with the host binding");
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_CHUNK   = ADAM_THREADS * 4 * 4;  // elements per 256-thread group and iteration (4 float4 per lane)

struct AdamCoef {
  float bc1, inv_sqrt_bc2, beta1, beta2, omb1, omb2, eps;
};
// hyper-parameters arrive as doubles and (1 - beta) is formed in double, as torch does: 1.0f - 0.999f is off by 1.3e-5
__device__ __forceinline__ AdamCoef adam_coefficients(double beta1d, double beta2d, float eps, const float* step_count) {
  const double t = (double) step_count[0] + 1.0;
  return AdamCoef{(float) (1.0 - pow(beta1d, t)), (float) (1.0 / sqrt(1.0 - pow(beta2d, t))), (float) beta1d, (float) beta2d,
      (float) (1.0 - beta1d), (float) (1.0 - beta2d), eps};
}

// which tensor owns a chunk: lane i keeps the first chunk of tensor i (loaded once by the caller into `first0`, INT64_MAX
// beyond the table); the owner is the number of tensors whose first chunk is <= chunk, minus one.  (A linear walk over the
// descriptors was a chain of dependent global loads per chunk -- ~30 of them for the tensors at the end of the table.)
__device__ __forceinline__ int adam_owner(const AdamTensor* __restrict__ tensors, int n_tensors, int64_t first0, int lane,
    int64_t chunk) {
  int ti = __popcll(__ballot(first0 <= chunk)) - 1;
  for (int base = 64; base < n_tensors; base += 64)  // (more than 64 tensors: rare)
    ti += __popcll(__ballot(base + lane < n_tensors && tensors[base + lane].chunk0 <= chunk));
  return __builtin_amdgcn_readfirstlane(ti);
}

// one chunk (ADAM_CHUNK elements from `base`) of tensor T, by 256 threads; t256 = this thread's index among them
__device__ __forceinline__ void adam_update_chunk(const AdamTensor& T, int64_t base, int t256, const AdamCoef& k) {
  const float step_size = T.lr / k.bc1;
  const bool aligned = ((reinterpret_cast<uintptr_t>(T.param) | reinterpret_cast<uintptr_t>(T.grad) |
                         reinterpret_cast<uintptr_t>(T.exp_avg) | reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15) == 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + ((int64_t) r * ADAM_THREADS + t256) * 4;
    if (aligned && i + 3 < T.n) {
      const float4 g = *reinterpret_cast<const float4*>(T.grad + i);
      float4 m = *reinterpret_cast<float4*>(T.exp_avg + i);
      float4 v = *reinterpret_cast<float4*>(T.exp_avg_sq + i);
      float4 p = *reinterpret_cast<float4*>(T.param + i);
#define SKGS_ADAM1(c)                                           \
  m.c = k.beta1 * m.c + k.omb1 * g.c;                           \
  v.c = k.beta2 * v.c + k.omb2 * g.c * g.c;                     \
  p.c -= step_size * m.c / (sqrtf(v.c) * k.inv_sqrt_bc2 + k.eps);
      SKGS_ADAM1(x) SKGS_ADAM1(y) SKGS_ADAM1(z) SKGS_ADAM1(w)
#undef SKGS_ADAM1
      *reinterpret_cast<float4*>(T.exp_avg + i)    = m;
      *reinterpret_cast<float4*>(T.exp_avg_sq + i) = v;
      *reinterpret_cast<float4*>(T.param + i)      = p;
    } else {
      for (int64_t e = i; e < T.n && e < i + 4; ++e) {
        const float g = T.grad[e];
        const float m = k.beta1 * T.exp_avg[e] + k.omb1 * g;
        const float v = k.beta2 * T.exp_avg_sq[e] + k.omb2 * g * g;
        T.exp_avg[e] = m, T.exp_avg_sq[e] = v;
        T.param[e] -= step_size * m / (sqrtf(v) * k.inv_sqrt_bc2 + k.eps);
      }
    }
  }
}

}  // namespace skgs
