// sp_backward.hip -- the skinning + weighting backward of the SUPERPOINT stage without atomics on the superpoint tables.
//
// With M = 512 superpoints the per-superpoint gradients of the skinning backward (14 values per (Gaussian, neighbour) pair:
// d spT (7), d d_rot (4), d d_scale (3)) and of the weighting backward (10: sp_hyper_feature (8), _sp_radius, _sp_weight) do
// not fit the dense moment contraction of the skeleton stage (deform.hip: M <= 64).  Scattering them with LDS float atomics
// is LDS-bound (a ds_add_f32 retires ~one LANE per 2-3 clocks: 7 M + 5 M lane-atomics = 58 + 68 us at P = 1e5,
// tools/pmc_kernel.sh); global atomics on 512 x 24 hot addresses are 8x worse (534 us measured).  So the reduction is turned
// around:
//   forward   the search (sp_knn.hip) also files every pair under its superpoint: inverse lists lists[M][cap] of pair ids
//             (n << 4 | k), slots reserved per workgroup (LDS histogram, ONE global atomic per touched superpoint and
//             workgroup: a few dozen when the Gaussians are in spatial order);
//   rows      one lane per Gaussian: everything of the skinning backward that belongs to the Gaussian (its four parameter
//             gradients, g_weights), then the weighting backward (softmax / kernel chain rule, hyper_feature.grad), and a
//             compact payload for the other pass: U[n] = [g_dx | g_v | g_ds] (10 floats) and V[n,k] = [g_dist, radius term,
//             kernel-weight term];
//   bones     SLICES workgroups per superpoint walk its list, gather p, U, the hyper row and V of every pair, form the 24
//             contributions in registers and tree-reduce them: plain stores of per-slice partials, no atomics;
//   finalize  adds the slices, applies d exp / d sigmoid of the raw radius / weight parameters, clears the list counters.
// Arithmetic of the contributions = deform.hip::deform_backward_kernel / sp_knn.hip::sp_weights_backward_kernel (same
// expressions, summed in list order instead of atomic order).
#include <algorithm>
#include <cstdint>

#include "skgs_common.h"
#include "deform_lane.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

constexpr int SB_THREADS = 256;
constexpr int MAXK       = SP_MAXK;
constexpr int MAXF       = SP_MAXF;
constexpr int UROW       = SP_UROW;  // per-Gaussian payload row (deform_lane.h)
constexpr int NV         = 24;   // per-superpoint sums: spT 7 | d_rot 4 | d_scale 3 | sp_hyper 8 | radius | kernel weight
constexpr int SLICES     = 4;    // workgroups per superpoint list

// ------------------------------------------------------------------------------------------------------------ rows
template <int F>
__global__ void __launch_bounds__(SB_THREADS) sp_backward_rows_kernel(int P, SpRowsArgs ja, const float* __restrict__ g_means,
    const float* __restrict__ g_scales, const float* __restrict__ g_rotations, const float* __restrict__ g_opacity) {
  extern __shared__ float s_bones[];  // [M][14]
  for (int j = threadIdx.x; j < ja.M; j += SB_THREADS) load_bone(ja.bone_T, ja.bone_drot, ja.bone_dscale, j, s_bones + j * BONE_F);
  __syncthreads();
  const int n = blockIdx.x * SB_THREADS + threadIdx.x;
  if (n >= P) return;
  const float g_dx[3] = {g_means[3 * n], g_means[3 * n + 1], g_means[3 * n + 2]};
  const float g_ds[3] = {g_scales[3 * n], g_scales[3 * n + 1], g_scales[3 * n + 2]};
  sp_rows_lane<F>(ja, s_bones, n, g_dx, g_ds, reinterpret_cast<const float4*>(g_rotations)[n], g_opacity[n]);
}

// ------------------------------------------------------------------------------------------------------------ bones
template <int F>
__global__ void __launch_bounds__(SB_THREADS) sp_backward_bones_kernel(int K, int M, int cap, const uint32_t* __restrict__ counts,
    const uint32_t* __restrict__ lists, const float* __restrict__ points, const float* __restrict__ weights,
    const float* __restrict__ bone_T, const float* __restrict__ bone_drot, const float* __restrict__ bone_dscale,
    const float* __restrict__ feature, const float* __restrict__ sp_feature, int logits, int largest, const float* __restrict__ U,
    const float* __restrict__ V, float* __restrict__ partials) {
  __shared__ float s_red[SB_THREADS / 64][NV];
  const int j = blockIdx.x / SLICES, slice = blockIdx.x % SLICES;
  const int cnt = (int) min(counts[j], (uint32_t) cap);
  float b[BONE_F];
  load_bone(bone_T, bone_drot, bone_dscale, j, b);
  const float q0 = bone_T[7 * j + 3], q1 = bone_T[7 * j + 4], q2 = bone_T[7 * j + 5], q3 = bone_T[7 * j + 6];
  const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);  // |q| of the raw quaternion (the row holds the unit one)
  float sf[MAXF];
#pragma unroll
  for (int c = 0; c < MAXF; ++c) sf[c] = (F > 0 && c < F) ? sp_feature[(size_t) j * F + c] : 0.f;
  float acc[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) acc[c] = 0.f;
  const uint32_t* L = lists + (size_t) j * cap;
  for (int e = slice * SB_THREADS + threadIdx.x; e < cnt; e += SLICES * SB_THREADS) {
    const uint32_t pid = L[e];
    const int n = (int) (pid >> 4), k = (int) (pid & 15u);
    const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
    const float w    = weights[(size_t) n * K + k];
    // warp_method `largest`: the position's gradient reaches only the bone of the Gaussian's largest weight, unweighted
    const float wx   = largest ? (argmax_slot(weights + (size_t) n * K, K) == k ? 1.f : 0.f) : w;
    const float4 u0 = reinterpret_cast<const float4*>(U + (size_t) n * UROW)[0];
    const float4 u1 = reinterpret_cast<const float4*>(U + (size_t) n * UROW)[1];
    const float4 u2 = reinterpret_cast<const float4*>(U + (size_t) n * UROW)[2];
    const float g_v[4]  = {u0.w, u1.x, u1.y, u1.z};
    const float g_ds[3] = {u1.w, u2.x, u2.y};
    // d spT: translation = w g_dx; quaternion through R(q) p and the normalisation (deform.hip::deform_backward_kernel)
    const float g[3] = {wx * u0.x, wx * u0.y, wx * u0.z};
    const float* vq  = b;
    const float vxp[3] = {vq[1] * p[2] - vq[2] * p[1], vq[2] * p[0] - vq[0] * p[2], vq[0] * p[1] - vq[1] * p[0]};
    const float pxg[3] = {p[1] * g[2] - p[2] * g[1], p[2] * g[0] - p[0] * g[2], p[0] * g[1] - p[1] * g[0]};
    const float vdp = vq[0] * p[0] + vq[1] * p[1] + vq[2] * p[2];
    const float gdv = g[0] * vq[0] + g[1] * vq[1] + g[2] * vq[2];
    const float gdp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
    float gqh[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) gqh[c] = 2.0f * vq[3] * pxg[c] + 2.0f * (vdp * g[c] + gdv * p[c] - 2.0f * gdp * vq[c]);
    gqh[3] = 2.0f * (g[0] * vxp[0] + g[1] * vxp[1] + g[2] * vxp[2]);
    const float dotq = vq[0] * gqh[0] + vq[1] * gqh[1] + vq[2] * gqh[2] + vq[3] * gqh[3];
    acc[0] += g[0], acc[1] += g[1], acc[2] += g[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[3 + c] += (gqh[c] - vq[c] * dotq) / qn;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[7 + c] += w * g_v[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[11 + c] += w * g_ds[c];
    if (!logits) {
      const float4 vv = *reinterpret_cast<const float4*>(V + ((size_t) n * K + k) * 4);  // g_dist, radius term, kernel-weight term
      if (F > 0) {
        const float4 f0 = reinterpret_cast<const float4*>(feature + (size_t) n * F)[0];
        const float4 f1 = reinterpret_cast<const float4*>(feature + (size_t) n * F)[1];
        const float fc[MAXF] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
        for (int c = 0; c < MAXF; ++c) acc[14 + c] += -(vv.x * 2.f * (fc[c] - sf[c]));
      }
      acc[22] += vv.y;
      acc[23] += vv.z;
    }
  }
  // ---- 256 partial rows -> one: DPP sums per wave, the four waves through LDS
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const float t = wave_sum_to_lane63(acc[c]);
    if (lane == 63) s_red[wave][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < NV)
    partials[((size_t) j * SLICES + slice) * NV + threadIdx.x] =
        (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256) sp_backward_finalize_kernel(int M, int F, int logits, const float* __restrict__ partials,
    const float* __restrict__ radius_raw, const float* __restrict__ kweight_raw, float* __restrict__ g_bone_T,
    float* __restrict__ g_bone_drot, float* __restrict__ g_bone_dscale, float* __restrict__ g_sp_feature,
    float* __restrict__ g_radius, float* __restrict__ g_kweight, uint32_t* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * NV) return;
  const int j = i / NV, c = i % NV;
  float s = 0.f;
#pragma unroll
  for (int sl = 0; sl < SLICES; ++sl) s += partials[((size_t) j * SLICES + sl) * NV + c];
  if (c == 0) counts[j] = 0u;  // the lists are consumed: the next forward files its pairs from slot 0
  if (c < 7) {
    g_bone_T[7 * j + c] = s;
  } else if (c < 11) {
    g_bone_drot[4 * j + c - 7] = s;
  } else if (c < 14) {
    g_bone_dscale[3 * j + c - 11] = s;
  } else if (logits) {
    return;
  } else if (c < 22) {
    if (g_sp_feature && c - 14 < F) g_sp_feature[(size_t) j * F + c - 14] = s;
  } else if (c == 22) {
    if (g_radius) g_radius[j] = radius_raw ? s * expf(radius_raw[j]) : 0.f;  // d exp(x) = exp(x)
  } else if (g_kweight) {
    float d = 0.f;
    if (kweight_raw) {
      const float sg = 1.0f / (1.0f + expf(-kweight_raw[j]));
      d = sg * (1.f - sg);
    }
    g_kweight[j] = s * d;
  }
}

}  // namespace

// the inverse lists: [0,256) header {cap, overflow}, counts[M] (256-B aligned), lists[M][cap], the search's packed table [M][12]
size_t sp_pairs_bytes(int P, int M, int K) {
  const size_t cap = sp_pairs_capacity(P, M, K);
  return 256 + align256((size_t) M * 4) + (size_t) M * cap * 4 + align256((size_t) M * 48);
}

// ---- the call in pieces: argument checks, the rows launch (or the arguments of the rows job of the rasterizer's per-Gaussian
// backward launch, preprocess.hip), the bones + finalize launches
static size_t sp_skinning_workspace_bytes(int P, int M, int K) {
  if (P <= 0 || M <= 0 || K <= 0) return 0;
  return align256((size_t) P * UROW * 4) + align256((size_t) P * K * 16) + align256((size_t) M * SLICES * NV * 4) + 256;
}
struct SpSkinningScratch {
  float *U, *V, *partials;
};
static SpSkinningScratch sp_skinning_scratch(const skgs_sp_skinning_job& j) {
  const int P = std::max(j.in->P, 1), K = j.in->K;
  char* wsp = reinterpret_cast<char*>(j.workspace);
  return SpSkinningScratch{reinterpret_cast<float*>(wsp), reinterpret_cast<float*>(wsp + align256((size_t) P * UROW * 4)),
      reinterpret_cast<float*>(wsp + align256((size_t) P * UROW * 4) + align256((size_t) P * K * 16))};
}
int sp_skinning_check(const skgs_sp_skinning_job& j) {
  const skgs_deform_inputs* in = j.in;
  SKGS_REQUIRE(in && in->P >= 0 && in->K >= 1 && in->K <= MAXK && in->M >= 1, "sp_skinning_backward: need P >= 0, 1 <= K <= 16, M >= 1");
  SKGS_REQUIRE(j.F == 0 || j.F == 8, "sp_skinning_backward: F (hyper dimensions) must be 0 or 8");
  const int P = in->P, K = in->K, M = in->M;
  SKGS_REQUIRE(in->live_count == nullptr, "sp_skinning_backward: no row capacity in stage sp");
  SKGS_REQUIRE(j.pairs && j.pairs_bytes >= sp_pairs_bytes(std::max(P, 1), M, K), "sp_skinning_backward: pair lists missing or too small");
  SKGS_REQUIRE(j.workspace && j.workspace_bytes >= sp_skinning_workspace_bytes(std::max(P, 1), M, K),
      "sp_skinning_backward: workspace too small");
  SKGS_REQUIRE(j.g_bone_T && j.g_bone_drot && j.g_bone_dscale, "sp_skinning_backward: superpoint gradient outputs are required");
  SKGS_REQUIRE(P == 0 || (in->points && in->weights && in->indices && in->bone_T && in->bone_drot && in->bone_dscale &&
                             in->log_scale && in->rot && in->opacity_logit && j.g_xyz && j.g_log_scale && j.g_rot && j.g_opacity_logit),
      "sp_skinning_backward: NULL argument");
  SKGS_REQUIRE(j.logit_weighting || P == 0 || (j.nn_dist && (j.F == 0 || (j.feature && j.sp_feature))),
      "sp_skinning_backward: the weighting's inputs are required");
  SKGS_REQUIRE((size_t) M * BONE_F * 4 <= 64 * 1024, "sp_skinning_backward: too many superpoints for the LDS table");
  return 0;
}
static int sp_skinning_rest_launches(const skgs_sp_skinning_job& j, hipStream_t s) {
  const skgs_deform_inputs* in = j.in;
  const int P = in->P, K = in->K, M = in->M, F = j.F;
  const SpSkinningScratch w = sp_skinning_scratch(j);
  SpPairsView pv            = sp_pairs_view(j.pairs, std::max(P, 1), M, K);
#define SKGS_BONES(F_)                                                                                                          \
  hipLaunchKernelGGL((sp_backward_bones_kernel<F_>), dim3(M * SLICES), dim3(SB_THREADS), 0, s, K, M, pv.cap, pv.counts, pv.lists, \
      in->points, in->weights, in->bone_T, in->bone_drot, in->bone_dscale, j.feature, j.sp_feature, (int) j.logit_weighting, \
      in->largest ? 1 : 0, w.U, w.V, w.partials)
  if (F == 8) SKGS_BONES(8); else SKGS_BONES(0);
#undef SKGS_BONES
  SKGS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sp_backward_finalize_kernel, dim3((M * NV + 255) / 256), dim3(256), 0, s, M, F, (int) j.logit_weighting, w.partials,
      j.sp_radius_raw, j.sp_weight_raw, j.g_bone_T, j.g_bone_drot, j.g_bone_dscale, j.g_sp_feature, j.g_sp_radius, j.g_sp_weight,
      pv.counts);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sp_skinning_rest(const skgs_sp_skinning_job& j, hipStream_t s) {  // (the rows pass ran as the rasterizer's job)
  ProfScope prof(K_DEFORM_BWD, s);
  return sp_skinning_rest_launches(j, s);
}

}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_sp_pairs_bytes(int32_t P, int32_t M, int32_t K) { return (P > 0 && M > 0 && K > 0) ? sp_pairs_bytes(P, M, K) : 0; }
size_t skgs_sp_skinning_backward_workspace_bytes(int32_t P, int32_t M, int32_t K) { return sp_skinning_workspace_bytes(P, M, K); }

int skgs_sp_skinning_backward(const skgs_deform_inputs* in, int32_t F, const float* feature, const float* sp_feature,
    const float* sp_radius_raw, const float* sp_weight_raw, float temperature, int32_t logit_weighting, const float* nn_dist,
    const float* g_means, const float* g_scales, const float* g_rotations, const float* g_opacity, float* g_weights, float* g_xyz,
    float* g_log_scale, float* g_rot, float* g_opacity_logit, float* g_feature, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_sp_feature, float* g_sp_radius, float* g_sp_weight, void* pairs, size_t pairs_bytes,
    void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  const skgs_sp_skinning_job j{in, F, feature, sp_feature, sp_radius_raw, sp_weight_raw, temperature, logit_weighting, nn_dist,
      g_weights, g_xyz, g_log_scale, g_rot, g_opacity_logit, g_feature, g_bone_T, g_bone_drot, g_bone_dscale, g_sp_feature, g_sp_radius,
      g_sp_weight, pairs, pairs_bytes, workspace, workspace_bytes};
  if (sp_skinning_check(j)) return 1;
  const int P = in->P;
  SKGS_REQUIRE(P == 0 || (g_means && g_scales && g_rotations && g_opacity), "sp_skinning_backward: NULL argument");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_DEFORM_BWD, s);
  if (P > 0) {
    const SpRowsArgs ra = sp_rows_args(j);
    const dim3 grid((P + SB_THREADS - 1) / SB_THREADS), block(SB_THREADS);
    if (F == 8)
      hipLaunchKernelGGL((sp_backward_rows_kernel<8>), grid, block, sp_rows_lds_bytes(in->M), s, P, ra, g_means, g_scales,
          g_rotations, g_opacity);
    else
      hipLaunchKernelGGL((sp_backward_rows_kernel<0>), grid, block, sp_rows_lds_bytes(in->M), s, P, ra, g_means, g_scales,
          g_rotations, g_opacity);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  return sp_skinning_rest_launches(j, s);
}

}  // extern "C"
