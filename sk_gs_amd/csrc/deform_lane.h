// deform_lane.h -- per-lane pieces of the skeleton-stage deform (K nearest bones, softmax weights, linear-blend skinning,
// activations: networks/sk_gs.py:757-770,1143-1150,1162,1192-1203; SE3 semantics my_ext/_C/include/lie.h:45-64,246), shared by
// deform.hip (the deform launches) and preprocess.hip (the launch that runs the deform in front of the rasterizer's per-Gaussian
// pass).  Both translation units compile with `fp contract(off)`: the same expressions give the same bits in either.
#pragma once
#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

constexpr int BONE_F = 14;  // qx qy qz qw tx ty tz | drot[4] | dscale[3]

__device__ __forceinline__ void load_bone(const float* T7, const float* drot, const float* dscale, int j, float* b) {
  const float q0 = T7[7 * j + 3], q1 = T7[7 * j + 4], q2 = T7[7 * j + 5], q3 = T7[7 * j + 6];
  const float n  = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  b[0] = q0 / n, b[1] = q1 / n, b[2] = q2 / n, b[3] = q3 / n;
  b[4] = T7[7 * j], b[5] = T7[7 * j + 1], b[6] = T7[7 * j + 2];
  b[7] = drot[4 * j], b[8] = drot[4 * j + 1], b[9] = drot[4 * j + 2], b[10] = drot[4 * j + 3];
  b[11] = dscale[3 * j], b[12] = dscale[3 * j + 1], b[13] = dscale[3 * j + 2];
}

// y = p + w*uv + q x uv + t, uv = 2 q x p   (lie.h:59-64,246)
__device__ __forceinline__ void se3_act(const float* b, const float* p, float* y) {
  float uv[3] = {b[1] * p[2] - b[2] * p[1], b[2] * p[0] - b[0] * p[2], b[0] * p[1] - b[1] * p[0]};
  uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
  const float c[3] = {b[1] * uv[2] - b[2] * uv[1], b[2] * uv[0] - b[0] * uv[2], b[0] * uv[1] - b[1] * uv[0]};
  y[0] = p[0] + b[3] * uv[0] + c[0] + b[4];
  y[1] = p[1] + b[3] * uv[1] + c[1] + b[5];
  y[2] = p[2] + b[3] * uv[2] + c[2] + b[6];
}

// KCAP = compile-time capacity of the per-lane top-K list (>= K): the insertion network is KCAP steps per bone.
// One bubble step per slot as selects (v_cndmask), no branches and no array copies: a candidate displaces the first entry it
// is strictly smaller than and the displaced entry moves on, so equal distances stay behind earlier (lower) indices.  Slots
// beyond K just collect the overflow; the first K are the top-K.
template <int KCAP>
__device__ __forceinline__ void topk_insert(float (&bd)[KCAP], int (&bi)[KCAP], float cd, int ci) {
#pragma unroll
  for (int k = 0; k < KCAP; ++k) {
    const bool lt  = cd < bd[k];
    const float td = bd[k];
    const int ti   = bi[k];
    bd[k] = lt ? cd : td;
    bi[k] = lt ? ci : ti;
    cd    = lt ? td : cd;
    ci    = lt ? ti : ci;
  }
}

// One Gaussian of knn_weights_kernel + deform_forward_kernel: the K nearest of the M joints in LDS (squared L2, ascending,
// ties -> lower index), softmax over the K logits `logit(bone)` of this Gaussian's row of sp_W, and the three blended sums
// sx = sum w T_i(p), sr = sum w d_rot_i, ss = sum w d_scale_i.  w[k] / bi[k] (k < K) are the weights and bone ids.
template <int KCAP, class Logit>
__device__ __forceinline__ void knn_softmax_skin_lane(int M, int K, const float* s_j, const float* s_bones, const float (&p)[3],
    Logit logit, float (&w)[KCAP], int (&bi)[KCAP], float (&sx)[3], float (&sr)[4], float (&ss)[3]) {
  float bd[KCAP];
#pragma unroll
  for (int k = 0; k < KCAP; ++k) bd[k] = __builtin_inff(), bi[k] = 0;
  for (int j = 0; j < M; ++j) {
    const float d0 = p[0] - s_j[3 * j], d1 = p[1] - s_j[3 * j + 1], d2 = p[2] - s_j[3 * j + 2];
    float d = 0.f;
    d += d0 * d0;
    d += d1 * d1;
    d += d2 * d2;
    topk_insert<KCAP>(bd, bi, d, j);
  }
  float l[KCAP];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < KCAP; ++k) {
    l[k] = k < K ? logit(bi[k]) : -INFINITY;
    mx   = fmaxf(mx, l[k]);
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KCAP; ++k) {
    l[k] = k < K ? expf(l[k] - mx) : 0.f;
    sum += l[k];
  }
  sx[0] = sx[1] = sx[2] = 0.f, sr[0] = sr[1] = sr[2] = sr[3] = 0.f, ss[0] = ss[1] = ss[2] = 0.f;
#pragma unroll
  for (int k = 0; k < KCAP; ++k) {
    w[k] = 0.f;
    if (k < K) {
      w[k] = l[k] / sum;
      const float* b = s_bones + bi[k] * BONE_F;
      float y[3];
      se3_act(b, p, y);
      sx[0] += y[0] * w[k], sx[1] += y[1] * w[k], sx[2] += y[2] * w[k];
      sr[0] += b[7] * w[k], sr[1] += b[8] * w[k], sr[2] += b[9] * w[k], sr[3] += b[10] * w[k];
      ss[0] += b[11] * w[k], ss[1] += b[12] * w[k], ss[2] += b[13] * w[k];
    }
  }
}

// the activation epilogue of one Gaussian (sk_gs.py:1162,1192,1202-1203): means = xyz + (sx - p), scales = exp(log_scale) + ss,
// rotation = normalize(rot + sr) (eps 1e-12), opacity = sigmoid(logit)
__device__ __forceinline__ void deform_activate_lane(const float (&p)[3], const float (&sx)[3], const float (&sr)[4],
    const float (&ss)[3], const float (&xyz)[3], const float (&log_scale)[3], float4 r4, float opacity_logit, float (&means)[3],
    float (&scales)[3], float4& rotation, float& opacity) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dx = sx[c] - p[c];
    means[c]  = xyz[c] + dx;
    scales[c] = expf(log_scale[c]) + ss[c];
  }
  const float v[4] = {r4.x + sr[0], r4.y + sr[1], r4.z + sr[2], r4.w + sr[3]};
  float nv = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  nv       = fmaxf(nv, 1e-12f);
  rotation = make_float4(v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv);
  opacity  = 1.0f / (1.0f + expf(-opacity_logit));
}

}  // namespace
}  // namespace skgs
