// deform.hip -- skeleton / superpoint linear-blend-skinning deform fused with the activation epilogue (gfx950).
//
// Reference op sequence (networks/sk_gs.py:1143-1150,1162,1192-1203, SE3 semantics my_ext/_C/include/lie.h:45-64,246):
//   ~10 torch/lietorch kernels that materialise [P,K,7] gathered transforms and [P,K,3] warped points, then four
//   element-wise activation kernels.  Here: ONE streaming kernel per direction, one lane per Gaussian; the bone
//   table (M x 14 floats: unit quaternion, translation, d_rot, d_scale) lives in LDS; no [P,K,*] temporary exists.
//   Backward scatter-adds bone gradients into an LDS copy of the bone table (ds_add_f32) and flushes it with one
//   global atomic per (bone, component, workgroup).
// HBM-bound: ~(88 + 12K) B per Gaussian forward, arithmetic in the oracle's order without contraction.
#include <algorithm>

#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

constexpr int DEFORM_THREADS = 256;
constexpr int BONE_F         = 14;    // qx qy qz qw tx ty tz | drot[4] | dscale[3]
constexpr int PREF_K         = 8;     // neighbour slots prefetched into registers (K is 5 in every shipped config)
constexpr int MAX_LDS_BONES  = 1024;  // 56 KB of dynamic LDS (backward keeps a gradient copy too: 512 bones)

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

template <bool LDS_BONES>
__global__ void __launch_bounds__(DEFORM_THREADS) deform_forward_kernel(int P, int K, int M, const float* __restrict__ points,
    const float* __restrict__ weights, const int64_t* __restrict__ indices, const float* __restrict__ bone_T,
    const float* __restrict__ bone_drot, const float* __restrict__ bone_dscale, const float* __restrict__ xyz,
    const float* __restrict__ log_scale, const float* __restrict__ rot, const float* __restrict__ opacity_logit,
    float* __restrict__ means, float* __restrict__ scales, float* __restrict__ rotations, float* __restrict__ opacity,
    float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale) {
  extern __shared__ float s_bones[];
  if (LDS_BONES) {
    for (int j = threadIdx.x; j < M; j += DEFORM_THREADS) load_bone(bone_T, bone_drot, bone_dscale, j, s_bones + j * BONE_F);
    __syncthreads();
  }
  const int n = blockIdx.x * DEFORM_THREADS + threadIdx.x;
  if (n >= P) return;
  const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
  float sx[3] = {0, 0, 0}, sr[4] = {0, 0, 0, 0}, ss[3] = {0, 0, 0};
  for (int k = 0; k < K; ++k) {
    const int j   = (int) indices[(size_t) n * K + k];
    const float w = weights[(size_t) n * K + k];
    float bl[BONE_F];
    const float* b;
    if (LDS_BONES) {
      b = s_bones + j * BONE_F;
    } else {
      load_bone(bone_T, bone_drot, bone_dscale, j, bl);
      b = bl;
    }
    float y[3];
    se3_act(b, p, y);
    sx[0] += y[0] * w, sx[1] += y[1] * w, sx[2] += y[2] * w;
    sr[0] += b[7] * w, sr[1] += b[8] * w, sr[2] += b[9] * w, sr[3] += b[10] * w;
    ss[0] += b[11] * w, ss[1] += b[12] * w, ss[2] += b[13] * w;
  }
  float v[4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dx = sx[c] - p[c];
    if (d_xyz) d_xyz[3 * n + c] = dx;
    means[3 * n + c]  = xyz[3 * n + c] + dx;
    scales[3 * n + c] = expf(log_scale[3 * n + c]) + ss[c];
    if (d_scale) d_scale[3 * n + c] = ss[c];
  }
  const float4 r4 = reinterpret_cast<const float4*>(rot)[n];
  v[0] = r4.x + sr[0], v[1] = r4.y + sr[1], v[2] = r4.z + sr[2], v[3] = r4.w + sr[3];
  if (d_rot) reinterpret_cast<float4*>(d_rot)[n] = make_float4(sr[0], sr[1], sr[2], sr[3]);
  float nv = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  nv       = fmaxf(nv, 1e-12f);
  reinterpret_cast<float4*>(rotations)[n] = make_float4(v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv);
  opacity[n] = 1.0f / (1.0f + expf(-opacity_logit[n]));
}

// Bone gradient rows (LDS and flush): 0..2 dT.t, 3..6 dT.q, 7..10 d_rot, 11..13 d_scale
template <bool LDS_BONES>
__global__ void __launch_bounds__(DEFORM_THREADS) deform_backward_kernel(int P, int K, int M, const float* __restrict__ points,
    const float* __restrict__ weights, const int64_t* __restrict__ indices, const float* __restrict__ bone_T,
    const float* __restrict__ bone_drot, const float* __restrict__ bone_dscale, const float* __restrict__ log_scale,
    const float* __restrict__ rot, const float* __restrict__ opacity_logit, const float* __restrict__ g_means,
    const float* __restrict__ g_scales, const float* __restrict__ g_rotations, const float* __restrict__ g_opacity,
    float* __restrict__ g_weights, float* __restrict__ g_bone_T, float* __restrict__ g_bone_drot,
    float* __restrict__ g_bone_dscale, float* __restrict__ g_xyz, float* __restrict__ g_log_scale,
    float* __restrict__ g_rot, float* __restrict__ g_opacity_logit, int ncopy) {
  extern __shared__ float s_mem[];
  float* s_bones = s_mem;                                   // [M][14] (LDS_BONES only)
  float* s_grad  = s_mem + (LDS_BONES ? M * BONE_F : 0);    // [ncopy][M][14] (LDS_BONES only)
  // With few bones, neighbouring lanes hit the same LDS row and ds_add_f32 serialises per conflicting lane (measured
  // ~5 cycles per lane-atomic at M = 20).  ncopy (power of two) private copies of the table, chosen by lane id, cut the
  // conflict degree by ncopy; they are summed at the flush.
  float* s_my = s_grad + (size_t) (threadIdx.x & (ncopy - 1)) * M * BONE_F;
  if (LDS_BONES) {
    for (int j = threadIdx.x; j < M; j += DEFORM_THREADS) load_bone(bone_T, bone_drot, bone_dscale, j, s_bones + j * BONE_F);
    for (int i = threadIdx.x; i < ncopy * M * BONE_F; i += DEFORM_THREADS) s_grad[i] = 0.f;
    __syncthreads();
  }
  for (int n = blockIdx.x * DEFORM_THREADS + threadIdx.x; n < P; n += gridDim.x * DEFORM_THREADS) {
    const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
    // All neighbour ids / weights are fetched up front (static registers, loads in flight together): with ~1.5 waves
    // per SIMD at P = 1e5 a load-use chain per k was pure HBM latency (measured 30 us for this loop alone).
    int jj[PREF_K];
    float ww[PREF_K];
#pragma unroll
    for (int k = 0; k < PREF_K; ++k) {
      jj[k] = k < K ? (int) indices[(size_t) n * K + k] : 0;
      ww[k] = k < K ? weights[(size_t) n * K + k] : 0.f;
    }
    float sr[4] = {0, 0, 0, 0};
    auto acc_sr = [&](int j, float w) {
      if (LDS_BONES) {
        const float* b = s_bones + j * BONE_F;
        sr[0] += b[7] * w, sr[1] += b[8] * w, sr[2] += b[9] * w, sr[3] += b[10] * w;
      } else {
        sr[0] += bone_drot[4 * j] * w, sr[1] += bone_drot[4 * j + 1] * w, sr[2] += bone_drot[4 * j + 2] * w,
            sr[3] += bone_drot[4 * j + 3] * w;
      }
    };
#pragma unroll
    for (int k = 0; k < PREF_K; ++k)
      if (k < K) acc_sr(jj[k], ww[k]);
    for (int k = PREF_K; k < K; ++k) acc_sr((int) indices[(size_t) n * K + k], weights[(size_t) n * K + k]);
    const float4 r4  = reinterpret_cast<const float4*>(rot)[n];
    const float4 gr4 = reinterpret_cast<const float4*>(g_rotations)[n];
    const float v[4] = {r4.x + sr[0], r4.y + sr[1], r4.z + sr[2], r4.w + sr[3]};
    const float gr[4] = {gr4.x, gr4.y, gr4.z, gr4.w};
    const float nv    = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    float g_v[4];
    if (nv > 1e-12f) {
      const float u[4] = {v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv};
      const float dot  = u[0] * gr[0] + u[1] * gr[1] + u[2] * gr[2] + u[3] * gr[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) g_v[c] = (gr[c] - u[c] * dot) / nv;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) g_v[c] = gr[c] / 1e-12f;
    }
    const float g_dx[3] = {g_means[3 * n], g_means[3 * n + 1], g_means[3 * n + 2]};
    const float g_ds[3] = {g_scales[3 * n], g_scales[3 * n + 1], g_scales[3 * n + 2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g_xyz[3 * n + c]       = g_dx[c];
      g_log_scale[3 * n + c] = g_ds[c] * expf(log_scale[3 * n + c]);
    }
    reinterpret_cast<float4*>(g_rot)[n] = make_float4(g_v[0], g_v[1], g_v[2], g_v[3]);
    const float sg     = 1.0f / (1.0f + expf(-opacity_logit[n]));
    g_opacity_logit[n] = g_opacity[n] * sg * (1.0f - sg);
    auto bone_body = [&](int k, int j, float w) {
      float bl[BONE_F];
      const float* b;
      if (LDS_BONES) {
        b = s_bones + j * BONE_F;
      } else {
        load_bone(bone_T, bone_drot, bone_dscale, j, bl);
        b = bl;
      }
      float y[3];
      se3_act(b, p, y);
      float gw = g_dx[0] * y[0] + g_dx[1] * y[1] + g_dx[2] * y[2];
#pragma unroll
      for (int c = 0; c < 4; ++c) gw += g_v[c] * b[7 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) gw += g_ds[c] * b[11 + c];
      g_weights[(size_t) n * K + k] = gw;
      float out[BONE_F];
      const float g[3] = {w * g_dx[0], w * g_dx[1], w * g_dx[2]};
      out[0] = g[0], out[1] = g[1], out[2] = g[2];
      const float* vq    = b;  // unit quaternion (x,y,z,w)
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
      // |q| of the raw bone quaternion (the table holds the unit one)
      const float q0 = bone_T[7 * j + 3], q1 = bone_T[7 * j + 4], q2 = bone_T[7 * j + 5], q3 = bone_T[7 * j + 6];
      const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
#pragma unroll
      for (int c = 0; c < 4; ++c) out[3 + c] = (gqh[c] - vq[c] * dotq) / qn;
#pragma unroll
      for (int c = 0; c < 4; ++c) out[7 + c] = w * g_v[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) out[11 + c] = w * g_ds[c];
      if (LDS_BONES) {
#pragma unroll
        for (int c = 0; c < BONE_F; ++c) atomicAdd(&s_my[j * BONE_F + c], out[c]);
      } else {
#pragma unroll
        for (int c = 0; c < 7; ++c) atomicAdd(&g_bone_T[7 * j + c], out[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(&g_bone_drot[4 * j + c], out[7 + c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&g_bone_dscale[3 * j + c], out[11 + c]);
      }
    };
#pragma unroll
    for (int k = 0; k < PREF_K; ++k)
      if (k < K) bone_body(k, jj[k], ww[k]);
    for (int k = PREF_K; k < K; ++k) bone_body(k, (int) indices[(size_t) n * K + k], weights[(size_t) n * K + k]);
  }
  if (LDS_BONES) {
    __syncthreads();
    for (int i = threadIdx.x; i < M * BONE_F; i += DEFORM_THREADS) {
      float val = 0.f;
      for (int cpy = 0; cpy < ncopy; ++cpy) val += s_grad[(size_t) cpy * M * BONE_F + i];
      if (val != 0.f) {
        const int j = i / BONE_F, c = i % BONE_F;
        if (c < 7)
          atomicAdd(&g_bone_T[7 * j + c], val);
        else if (c < 11)
          atomicAdd(&g_bone_drot[4 * j + c - 7], val);
        else
          atomicAdd(&g_bone_dscale[3 * j + c - 11], val);
      }
    }
  }
}

// K nearest bones (squared L2, ascending, ties -> lower index). joints staged in LDS.
constexpr int KNN_MAXK = 16;
// KCAP = compile-time capacity of the per-lane top-K list (>= K): the insertion network is KCAP steps per bone
template <int KCAP>
__global__ void __launch_bounds__(256) knn_bones_kernel(int P, int M, int K, int dim, const float* __restrict__ points,
    const float* __restrict__ joints, float* __restrict__ out_dist, int64_t* __restrict__ out_idx, int lds_joints) {
  extern __shared__ float s_j[];
  if (lds_joints) {
    for (int i = threadIdx.x; i < M * dim; i += blockDim.x) s_j[i] = joints[i];
    __syncthreads();
  }
  const float* jt = lds_joints ? s_j : joints;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P) return;
  float bd[KCAP];
  int bi[KCAP];
#pragma unroll
  for (int k = 0; k < KCAP; ++k) bd[k] = __builtin_inff(), bi[k] = -1;
  const float* pn = points + (size_t) n * dim;
  const float p0 = pn[0], p1 = dim > 1 ? pn[1] : 0.f, p2 = dim > 2 ? pn[2] : 0.f;
  for (int j = 0; j < M; ++j) {
    float d = 0.f;
    if (dim == 3) {  // the sk stage: xyz only (no runtime-indexed per-thread array: those live in scratch memory)
      const float d0 = p0 - jt[3 * j], d1 = p1 - jt[3 * j + 1], d2 = p2 - jt[3 * j + 2];
      d += d0 * d0;
      d += d1 * d1;
      d += d2 * d2;
    } else {
      for (int c = 0; c < dim; ++c) {
        const float df = pn[c] - jt[(size_t) j * dim + c];
        d += df * df;
      }
    }
    // insert (d, j) keeping ascending order; equal distances stay behind earlier (lower) indices
    float cd = d;
    int ci   = j;
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      if (k < K && cd < bd[k]) {
        const float td = bd[k];
        const int ti   = bi[k];
        bd[k] = cd, bi[k] = ci;
        cd = td, ci = ti;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KCAP; ++k)
    if (k < K) {
      out_dist[(size_t) n * K + k] = bi[k] >= 0 ? bd[k] : 0.f;
      out_idx[(size_t) n * K + k]  = bi[k];
    }
}

}  // namespace

int launch_deform_forward(const skgs_deform_inputs& in, float* means, float* scales, float* rotations, float* opacity,
    float* d_xyz, float* d_rot, float* d_scale, hipStream_t s) {
  if (in.P == 0) return 0;
  ProfScope prof(K_DEFORM_FWD, s);
  dim3 grid((in.P + DEFORM_THREADS - 1) / DEFORM_THREADS), block(DEFORM_THREADS);
  if (in.M <= MAX_LDS_BONES)
    hipLaunchKernelGGL(deform_forward_kernel<true>, grid, block, (size_t) in.M * BONE_F * 4, s, in.P, in.K, in.M, in.points,
        in.weights, in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.xyz, in.log_scale, in.rot, in.opacity_logit,
        means, scales, rotations, opacity, d_xyz, d_rot, d_scale);
  else
    hipLaunchKernelGGL(deform_forward_kernel<false>, grid, block, 0, s, in.P, in.K, in.M, in.points, in.weights,
        in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.xyz, in.log_scale, in.rot, in.opacity_logit, means, scales,
        rotations, opacity, d_xyz, d_rot, d_scale);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_deform_backward(const skgs_deform_inputs& in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, hipStream_t s) {
  if (in.P == 0) return 0;
  ProfScope prof(K_DEFORM_BWD, s);
  dim3 grid((in.P + DEFORM_THREADS - 1) / DEFORM_THREADS), block(DEFORM_THREADS);
  if (in.M <= MAX_LDS_BONES / 2) {
    const size_t table = (size_t) in.M * BONE_F * 4;
    int ncopy = 1;
    while (ncopy < 16 && table * (1 + 2 * ncopy) <= 56 * 1024) ncopy *= 2;
    hipLaunchKernelGGL(deform_backward_kernel<true>, grid, block, table * (1 + ncopy), s, in.P, in.K, in.M, in.points,
        in.weights, in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.log_scale, in.rot, in.opacity_logit, g_means,
        g_scales, g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot, g_bone_dscale, g_xyz, g_log_scale, g_rot,
        g_opacity_logit, ncopy);
  } else {
    hipLaunchKernelGGL(deform_backward_kernel<false>, grid, block, 0, s, in.P, in.K, in.M, in.points, in.weights,
        in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.log_scale, in.rot, in.opacity_logit, g_means, g_scales,
        g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot, g_bone_dscale, g_xyz, g_log_scale, g_rot,
        g_opacity_logit, 1);
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_knn_bones(int P, int M, int K, int dim, const float* points, const float* joints, float* out_dist,
    int64_t* out_idx, hipStream_t s) {
  if (P == 0) return 0;
  if (K > KNN_MAXK || K < 1) return set_error("knn_bones: K must be in [1,%d] (got %d)", KNN_MAXK, K);
  const size_t lds = (size_t) M * dim * 4;
  const int use_lds = lds <= 48 * 1024;
  ProfScope prof(K_KNN, s);
#define SKGS_KNN(KCAP_)                                                                                                  \
  hipLaunchKernelGGL(knn_bones_kernel<KCAP_>, dim3((P + 255) / 256), dim3(256), use_lds ? lds : 0, s, P, M, K, dim, points, \
      joints, out_dist, out_idx, use_lds)
  if (K <= 4)
    SKGS_KNN(4);
  else if (K <= 8)
    SKGS_KNN(8);
  else
    SKGS_KNN(KNN_MAXK);
#undef SKGS_KNN
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
