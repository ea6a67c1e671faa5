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
#include "deform_lane.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

constexpr int DEFORM_THREADS = 256;
constexpr int MAX_LDS_BONES  = 1024;  // 56 KB of dynamic LDS (backward keeps a gradient copy too: 512 bones)

template <bool LDS_BONES>
__global__ void __launch_bounds__(DEFORM_THREADS) deform_forward_kernel(int P, int K, int M, const float* __restrict__ points,
    const float* __restrict__ weights, const int64_t* __restrict__ indices, const float* __restrict__ bone_T,
    const float* __restrict__ bone_drot, const float* __restrict__ bone_dscale, const float* __restrict__ xyz,
    const float* __restrict__ log_scale, const float* __restrict__ rot, const float* __restrict__ opacity_logit,
    float* __restrict__ means, float* __restrict__ scales, float* __restrict__ rotations, float* __restrict__ opacity,
    float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale, int largest) {
  extern __shared__ float s_bones[];
  if (LDS_BONES) {
    for (int j = threadIdx.x; j < M; j += DEFORM_THREADS) load_bone(bone_T, bone_drot, bone_dscale, j, s_bones + j * BONE_F);
    __syncthreads();
  }
  const int n = blockIdx.x * DEFORM_THREADS + threadIdx.x;
  if (n >= P) return;
  const float p[3] = {points[3 * n], points[3 * n + 1], points[3 * n + 2]};
  float sx[3] = {0, 0, 0}, sr[4] = {0, 0, 0, 0}, ss[3] = {0, 0, 0};
  const int kmax = largest ? argmax_slot(weights + (size_t) n * K, K) : -1;  // warp_method `largest` (skgs_deform_inputs.largest)
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
    const float wx = kmax < 0 ? w : (k == kmax ? 1.f : 0.f);
    sx[0] += y[0] * wx, sx[1] += y[1] * wx, sx[2] += y[2] * wx;
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
    float* __restrict__ g_rot, float* __restrict__ g_opacity_logit, int ncopy, float* __restrict__ partials) {
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
  // (whole waves stay in the loop: the merged accumulation below needs all 64 lanes; `live` masks a wave's tail)
  for (int base = blockIdx.x * DEFORM_THREADS; base < P; base += gridDim.x * DEFORM_THREADS) {
    const bool live = base + (int) threadIdx.x < P;
    const int n     = live ? base + (int) threadIdx.x : P - 1;
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
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        g_xyz[3 * n + c]       = g_dx[c];
        g_log_scale[3 * n + c] = g_ds[c] * expf(log_scale[3 * n + c]);
      }
      reinterpret_cast<float4*>(g_rot)[n] = make_float4(g_v[0], g_v[1], g_v[2], g_v[3]);
      const float sg     = 1.0f / (1.0f + expf(-opacity_logit[n]));
      g_opacity_logit[n] = g_opacity[n] * sg * (1.0f - sg);
    }
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
      if (live) g_weights[(size_t) n * K + k] = gw;
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
        if (ncopy == 1) {  // many bones: merge the wave's lanes that name the same bone, then one 14-lane ds_add per bone
          wave_group_add<BONE_F>(s_grad, BONE_F, j, out, live);
        } else if (live) {
#pragma unroll
          for (int c = 0; c < BONE_F; ++c) atomicAdd(&s_my[j * BONE_F + c], out[c]);
        }
      } else if (live) {
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
      if (partials) {  // this workgroup's table as one contiguous partial: summed in workgroup order by the finalize launch
        partials[(size_t) blockIdx.x * M * BONE_F + i] = val;
      } else if (val != 0.f) {
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

// Many bones (the 512 superpoints of stage sp): every workgroup's [M][14] table used to be flushed with one global atomic per
// non-zero entry -- 391 workgroups x 7168 atomics onto the same 7168 addresses: 71 us at P = 1e5.  Now a fixed grid of
// workgroups walks the Gaussians, writes its table as a partial, and this launch sums the partials in workgroup order.
__global__ void __launch_bounds__(256) deform_backward_wide_finalize_kernel(int M, int nblk, const float* __restrict__ partials,
    float* __restrict__ g_bone_T, float* __restrict__ g_bone_drot, float* __restrict__ g_bone_dscale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * BONE_F) return;
  float sum = 0.f;
  for (int b0 = 0; b0 < nblk; b0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = partials[(size_t) min(b0 + u, nblk - 1) * M * BONE_F + i];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (b0 + u < nblk) sum += v[u];
  }
  const int j = i / BONE_F, c = i % BONE_F;
  if (c < 7)
    g_bone_T[7 * j + c] = sum;
  else if (c < 11)
    g_bone_drot[4 * j + c - 7] = sum;
  else
    g_bone_dscale[3 * j + c - 11] = sum;
}

// ------------------------------------------------------------------------------ bone gradients by moments
// Every bone gradient is linear in quantities that do not depend on the bone:
//     u_p = [ g_dx (3) | g_dx p^T (9, row-major S[a][b] = g_dx[a] p[b]) | g_v (4) | g_ds (3) ]          (19 floats)
//     Mom[b] = sum_{p,k : idx[p,k] = b} w[p,k] u_p            i.e.  Mom = W^T U  with the dense [P,M] weight matrix
//     g_T.t = Mom[0:3],  g_T.q = (I - q q^T) Q(q) Mom[3:12] / |q_raw|,  g_drot = Mom[12:16],  g_dscale = Mom[16:19]
// (Q(q) S is the quaternion gradient of sum w g.(R(q) p) written on the moment matrix, lie.h:59-64.)  So the scatter
// of 14 values per (Gaussian, neighbour) -- LDS atomics that serialise on the few bones neighbouring Gaussians share,
// 60 us at P = 1e5 -- becomes a small dense contraction: each wave stages its 64 weight rows and u vectors in LDS and
// every lane owns (bone, component) outputs, summing over the wave's Gaussians with conflict-free broadcast reads.
// Per-workgroup partial moments go to a workspace; a second tiny kernel reduces them in a fixed order and applies
// the per-bone linear maps.  No atomics: the bone gradients are deterministic.
__global__ void __launch_bounds__(DEFORM_BWD_THREADS) deform_backward_moments_kernel(int P, DeformBwdArgs a,
    const float* __restrict__ g_means, const float* __restrict__ g_scales, const float* __restrict__ g_rotations,
    const float* __restrict__ g_opacity, const int32_t* __restrict__ live_count /* NULL, or the live Gaussian count (<= P) */) {
  if (live_count) P = min(P, live_count[0]);  // the number of Gaussians is a device word: one captured graph survives densification
  if ((int) (blockIdx.x * DEFORM_BWD_THREADS) >= P) {
    deform_bwd_zero_partials(a);
    return;
  }
  extern __shared__ float s_mem[];
  const int n = blockIdx.x * DEFORM_BWD_THREADS + threadIdx.x;
  DeformBwdLane L;
  deform_bwd_prefetch(a, n, n < P, L);
  float g_dx[3] = {0, 0, 0}, g_ds[3] = {0, 0, 0}, go = 0.f;
  float4 gr4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) g_dx[c] = g_means[3 * n + c], g_ds[c] = g_scales[3 * n + c];
    gr4 = reinterpret_cast<const float4*>(g_rotations)[n], go = g_opacity[n];
  }
  deform_bwd_moments(a, P, s_mem, L, g_dx, g_ds, gr4, go);
}

// one workgroup per bone: fixed-order reduction of the per-workgroup partial moments, then the per-bone linear maps.
// Thread t owns the partial rows w = t, t + 256, ...: its 19 loads per row are independent (all in flight together;
// a per-component loop over w was a 49-deep dependent load chain, 15 us), then an LDS tree over the 256 threads.
__global__ void __launch_bounds__(256) deform_backward_finalize_kernel(int M, int nblk, const float* __restrict__ partials,
    const float* __restrict__ bone_T, float* __restrict__ g_bone_T, float* __restrict__ g_bone_drot,
    float* __restrict__ g_bone_dscale) {
  __shared__ float s_red[256][MOM_F + 1];
  const int b = blockIdx.x, t = threadIdx.x;
  float acc[MOM_F];
#pragma unroll
  for (int c = 0; c < MOM_F; ++c) acc[c] = 0.f;
  for (int w = t; w < nblk; w += 256) {
    const float* row = partials + ((size_t) w * M + b) * MOM_F;
#pragma unroll
    for (int c = 0; c < MOM_F; ++c) acc[c] += row[c];
  }
#pragma unroll
  for (int c = 0; c < MOM_F; ++c) s_red[t][c] = acc[c];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (t < d) {
#pragma unroll
      for (int c = 0; c < MOM_F; ++c) s_red[t][c] += s_red[t + d][c];
    }
    __syncthreads();
  }
  if (t != 0) return;
  float m[MOM_F];
  for (int i = 0; i < MOM_F; ++i) m[i] = s_red[0][i];
  const float q0 = bone_T[7 * b + 3], q1 = bone_T[7 * b + 4], q2 = bone_T[7 * b + 5], q3 = bone_T[7 * b + 6];
  const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const float vq[4] = {q0 / qn, q1 / qn, q2 / qn, q3 / qn};
  const float* S = m + 3;  // S[a][c] = sum w g[a] p[c]
#define SM(a, c) S[3 * (a) + (c)]
  const float pxg[3] = {SM(2, 1) - SM(1, 2), SM(0, 2) - SM(2, 0), SM(1, 0) - SM(0, 1)};  // sum p x g
  const float tr     = SM(0, 0) + SM(1, 1) + SM(2, 2);                                 // sum g . p
  float gqh[4];
  for (int i = 0; i < 3; ++i) {
    const float vdp_g = vq[0] * SM(i, 0) + vq[1] * SM(i, 1) + vq[2] * SM(i, 2);  // sum (q.p) g[i]
    const float gdv_p = vq[0] * SM(0, i) + vq[1] * SM(1, i) + vq[2] * SM(2, i);  // sum (g.q) p[i]
    gqh[i] = 2.0f * vq[3] * pxg[i] + 2.0f * (vdp_g + gdv_p - 2.0f * tr * vq[i]);
  }
  gqh[3] = 2.0f * ((vq[1] * SM(0, 2) - vq[2] * SM(0, 1)) + (vq[2] * SM(1, 0) - vq[0] * SM(1, 2)) +
                   (vq[0] * SM(2, 1) - vq[1] * SM(2, 0)));  // sum g . (q x p)
#undef SM
  const float dotq = vq[0] * gqh[0] + vq[1] * gqh[1] + vq[2] * gqh[2] + vq[3] * gqh[3];
  g_bone_T[7 * b] = m[0], g_bone_T[7 * b + 1] = m[1], g_bone_T[7 * b + 2] = m[2];
  for (int i = 0; i < 4; ++i) g_bone_T[7 * b + 3 + i] = (gqh[i] - vq[i] * dotq) / qn;
  for (int i = 0; i < 4; ++i) g_bone_drot[4 * b + i] = m[12 + i];
  for (int i = 0; i < 3; ++i) g_bone_dscale[3 * b + i] = m[16 + i];
}

// K nearest bones (squared L2, ascending, ties -> lower index). joints staged in LDS.
constexpr int KNN_MAXK = 16;
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
    topk_insert<KCAP>(bd, bi, cd, ci);
  }
#pragma unroll
  for (int k = 0; k < KCAP; ++k)
    if (k < K) {
      out_dist[(size_t) n * K + k] = bi[k] >= 0 ? bd[k] : 0.f;
      out_idx[(size_t) n * K + k]  = bi[k];
    }
}

// --------------------------------------------------------------------------------- LBS weights from logits
// weights = softmax_k( sp_W[p, indices[p, k]] )  -- the `sp_W` branch of calc_LBS_weight (sk_gs.py:769-770:
// torch.gather(sp_W, 1, indices).softmax(-1)); one lane per Gaussian, K <= 16.
__global__ void __launch_bounds__(256) lbs_weights_forward_kernel(int P, int M, int K, const float* __restrict__ sp_W,
    const int64_t* __restrict__ indices, float* __restrict__ weights) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float l[KNN_MAXK];
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    l[k] = sp_W[(size_t) p * M + (int) indices[(size_t) p * K + k]];
    mx   = fmaxf(mx, l[k]);
  }
  float sum = 0.f;
  for (int k = 0; k < K; ++k) {
    l[k] = expf(l[k] - mx);
    sum += l[k];
  }
  for (int k = 0; k < K; ++k) weights[(size_t) p * K + k] = l[k] / sum;
}

// g_logit[k] = w[k] * (g_w[k] - sum_j w[j] g_w[j]), scattered back through the gather into a DENSE row of g_sp_W
// (every element of the [P,M] gradient is written: zeros where the bone is not among the K nearest).  One lane per
// Gaussian builds its row in LDS; the workgroup then stores its 256 consecutive rows as one contiguous span.
__global__ void __launch_bounds__(256) lbs_weights_backward_kernel(int P, int M, int K, const float* __restrict__ weights,
    const int64_t* __restrict__ indices, const float* __restrict__ g_weights, float* __restrict__ g_sp_W) {
  extern __shared__ float s_rows[];  // [256][M]
  const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
  float* row = s_rows + (size_t) threadIdx.x * M;
  for (int m = 0; m < M; ++m) row[m] = 0.f;
  if (p < P) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += weights[(size_t) p * K + k] * g_weights[(size_t) p * K + k];
    for (int k = 0; k < K; ++k) {
      const float w = weights[(size_t) p * K + k];
      row[(int) indices[(size_t) p * K + k]] += w * (g_weights[(size_t) p * K + k] - dot);  // KNN indices are distinct
    }
  }
  __syncthreads();
  const size_t n = (size_t) min(256, P - p0) * M;
  float* dst     = g_sp_W + (size_t) p0 * M;
  for (size_t i = threadIdx.x; i < n; i += 256) dst[i] = s_rows[i];
}

// Many bones (the 512 superpoints of the sp stage, networks/sk_gs.py:830-856; exps/default.yaml num_superpoints): a [256][M]
// row staging does not fit LDS.  One thread per four consecutive columns of a row: it writes the sum of the (at most K)
// logit gradients whose bone id falls on its columns, zeros otherwise -- the [P,M] gradient leaves as whole 16-byte
// stores, the K ids / gradients of a row are re-read from L1 by the M/4 threads that share it.
template <bool SOFTMAX_BACKWARD>
__global__ void __launch_bounds__(256) lbs_logits_dense_wide_kernel(int P, int M, int K, const float* __restrict__ weights,
    const int64_t* __restrict__ indices, const float* __restrict__ g_in /* g_weights, or g_logits */,
    float* __restrict__ g_sp_W) {
  const int M4       = (M + 3) >> 2;
  const long long t  = (long long) blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long) P * M4) return;
  const int p = (int) (t / M4), m0 = 4 * (int) (t - (long long) p * M4);
  // (every rounding spelled out: adam.hip::adam_logit_rows_kernel forms the same gradient on the fly and must agree to the bit)
  float dot = 0.f;
  if (SOFTMAX_BACKWARD)
    for (int k = 0; k < K; ++k) dot = __builtin_fmaf(weights[(size_t) p * K + k], g_in[(size_t) p * K + k], dot);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const int j = (int) indices[(size_t) p * K + k] - m0;
    if (j >= 0 && j < 4) {
      const float g = SOFTMAX_BACKWARD ? __fmul_rn(weights[(size_t) p * K + k], __fsub_rn(g_in[(size_t) p * K + k], dot))
                                       : g_in[(size_t) p * K + k];
      v[j] += g;
    }
  }
  float* dst = g_sp_W + (size_t) p * M + m0;
  if ((M & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    for (int c = 0; c < 4 && m0 + c < M; ++c) dst[c] = v[c];
  }
}

// The same backward in two halves, for view-parallel training: g_logits [P,K] (compact: what the ranks all-reduce,
// K/M of the dense size -- the KNN indices are identical on every rank) and its expansion into the dense [P,M] rows.
__global__ void __launch_bounds__(256) lbs_weights_backward_compact_kernel(int P, int K, const float* __restrict__ weights,
    const float* __restrict__ g_weights, float* __restrict__ g_logits) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float dot = 0.f;
  for (int k = 0; k < K; ++k) dot += weights[(size_t) p * K + k] * g_weights[(size_t) p * K + k];
  for (int k = 0; k < K; ++k) g_logits[(size_t) p * K + k] = weights[(size_t) p * K + k] * (g_weights[(size_t) p * K + k] - dot);
}
__global__ void __launch_bounds__(256) lbs_logits_scatter_kernel(int P, int M, int K, const int64_t* __restrict__ indices,
    const float* __restrict__ g_logits, float* __restrict__ g_sp_W) {
  extern __shared__ float s_rows[];  // [256][M]
  const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
  float* row = s_rows + (size_t) threadIdx.x * M;
  for (int m = 0; m < M; ++m) row[m] = 0.f;
  if (p < P)
    for (int k = 0; k < K; ++k) row[(int) indices[(size_t) p * K + k]] += g_logits[(size_t) p * K + k];
  __syncthreads();
  const size_t n = (size_t) min(256, P - p0) * M;
  float* dst     = g_sp_W + (size_t) p0 * M;
  for (size_t i = threadIdx.x; i < n; i += 256) dst[i] = s_rows[i];
}

// ------------------------------------------------------------- distance-based LBS weightings, fused with the search
// The two other branches of calc_LBS_weight (networks/sk_gs.py:757-770; `LBS_method` weighted_kernel is the class default,
// sk_gs.py:364, exps/d_nerf_sc_gs.yaml:31):
//   kernel / weighted_kernel   w = (exp(-d / (2 r_i^2)) [* s_i] + 1e-7) / sum_k (...)      r = kernel_radius, s = kernel_weight
//   dist                       w = softmax_k(-d / temperature)
// with d, i the squared distances / ids of the K nearest bones in `dim` dimensions (3 in stage sk; 3 + 8 hyper-feature
// dimensions in stage sp, sk_gs.py:753-755).  Round 2 ran these as ~8 element-wise torch launches + autograd on top of the
// KNN kernel; here one launch per direction.  The forward keeps the distances for the backward.
// kernel_radius / kernel_weight of bone j: the activated values, or (activate != 0) the raw parameters `_sp_radius` /
// `_sp_weight` run through their activations here -- exp and sigmoid, the properties of sk_gs.py:547-553 -- so that a step
// without autograd needs no launch for M values
__device__ __forceinline__ float dw_radius(const float* __restrict__ radius, int j, int activate) {
  return activate ? expf(radius[j]) : radius[j];
}
__device__ __forceinline__ float dw_kweight(const float* __restrict__ kweight, int j, int activate) {
  return activate ? 1.0f / (1.0f + expf(-kweight[j])) : kweight[j];
}
template <int KCAP>
__global__ void __launch_bounds__(256) knn_dist_weights_kernel(int P, int M, int K, int dim, const float* __restrict__ points,
    const float* __restrict__ joints, const float* __restrict__ radius, const float* __restrict__ kweight, float temperature,
    int64_t* __restrict__ out_idx, float* __restrict__ out_weights, float* __restrict__ out_dist, int lds_joints, int activate) {
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
  for (int k = 0; k < KCAP; ++k) bd[k] = __builtin_inff(), bi[k] = 0;
  const float* pn = points + (size_t) n * dim;
  const float p0 = pn[0], p1 = dim > 1 ? pn[1] : 0.f, p2 = dim > 2 ? pn[2] : 0.f;
  for (int j = 0; j < M; ++j) {  // the search of knn_bones_kernel: same sums, same order, same ties
    float d = 0.f;
    if (dim == 3) {
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
    topk_insert<KCAP>(bd, bi, d, j);
  }
  float v[KCAP];
  float sum = 0.f;
  if (radius) {  // sk_gs.py:760-766, in its order: exp(-d / (2 r^2)), * s, + 1e-7, / sum
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      v[k] = 0.f;
      if (k < K) {
        const float r = dw_radius(radius, bi[k], activate);
        float e = expf(-bd[k] / (2.f * (r * r)));
        if (kweight) e = e * dw_kweight(kweight, bi[k], activate);
        v[k] = e + 1e-7f;
        sum += v[k];
      }
    }
  } else {  // sk_gs.py:770: softmax(-d / temperature)
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      v[k] = k < K ? -bd[k] / temperature : -INFINITY;
      mx   = fmaxf(mx, v[k]);
    }
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      v[k] = k < K ? expf(v[k] - mx) : 0.f;
      sum += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < KCAP; ++k)
    if (k < K) {
      out_weights[(size_t) n * K + k] = v[k] / sum;
      out_idx[(size_t) n * K + k]     = bi[k];
      out_dist[(size_t) n * K + k]    = bd[k];
    }
}

// Backward of the above: g_w [P,K] -> g_d [P,K] (per method) -> g_points [P,dim] = sum_k 2 g_d (p - j_k) (optional), and per
// bone the sums  g_joints[j] = -sum 2 g_d (p - j),  g_radius[j] = sum g_e e d / r^3,  g_kweight[j] = sum g_v e  over the
// (Gaussian, neighbour) pairs that picked bone j (what autograd's index / gather backward scatter-add in the reference).
// A fixed grid of workgroups walks the Gaussians; each keeps [M][dim + 2] accumulators in LDS (ds_add_f32), writes them as
// one partial, and `dist_weights_finalize_kernel` adds the partials in workgroup order.
constexpr int DW_MAX_BLOCKS = 512;
__global__ void __launch_bounds__(256) dist_weights_backward_kernel(int P, int M, int K, int dim,
    const float* __restrict__ points, const float* __restrict__ joints, const float* __restrict__ radius,
    const float* __restrict__ kweight, float temperature, const float* __restrict__ weights,
    const int64_t* __restrict__ indices, const float* __restrict__ nn_dist, const float* __restrict__ g_weights,
    float* __restrict__ g_points, float* __restrict__ partials, int activate) {
  extern __shared__ float s_acc[];  // [M][V]
  const int V = dim + 2;
  for (int i = threadIdx.x; i < M * V; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  for (int n = blockIdx.x * 256 + threadIdx.x; n < P; n += gridDim.x * 256) {
    const float* w  = weights + (size_t) n * K;
    const float* gw = g_weights + (size_t) n * K;
    const float* dd = nn_dist + (size_t) n * K;
    const int64_t* ix = indices + (size_t) n * K;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += w[k] * gw[k];
    float sum = 0.f;
    if (radius)  // S = sum_k v_k is not stored: recompute it with the forward's arithmetic
      for (int k = 0; k < K; ++k) {
        const float r = dw_radius(radius, (int) ix[k], activate);
        float e = expf(-dd[k] / (2.f * (r * r)));
        if (kweight) e = e * dw_kweight(kweight, (int) ix[k], activate);
        sum += e + 1e-7f;
      }
    float g_d[KNN_MAXK];  // (compile-time indexed: stays in registers)
    int jj[KNN_MAXK];
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) {
      g_d[k] = 0.f, jj[k] = 0;
      if (k < K) {
        const int j = (int) ix[k];
        jj[k]       = j;
        if (radius) {
          const float r   = dw_radius(radius, j, activate);
          const float e   = expf(-dd[k] / (2.f * (r * r)));
          const float sk  = kweight ? dw_kweight(kweight, j, activate) : 1.f;
          const float g_v = (gw[k] - dot) / sum;
          const float g_e = g_v * sk;
          g_d[k]          = g_e * e * (-1.f / (2.f * (r * r)));
          atomicAdd(s_acc + (size_t) j * V + dim, g_e * e * (dd[k] / (r * r * r)));
          if (kweight) atomicAdd(s_acc + (size_t) j * V + dim + 1, g_v * e);
        } else {
          g_d[k] = -(w[k] * (gw[k] - dot)) / temperature;
        }
      }
    }
    for (int c = 0; c < dim; ++c) {
      const float pc = points[(size_t) n * dim + c];
      float gp = 0.f;
#pragma unroll
      for (int k = 0; k < KNN_MAXK; ++k)
        if (k < K) {
          const float t = g_d[k] * 2.f * (pc - joints[(size_t) jj[k] * dim + c]);
          gp += t;
          atomicAdd(s_acc + (size_t) jj[k] * V + c, -t);
        }
      if (g_points) g_points[(size_t) n * dim + c] = gp;
    }
  }
  __syncthreads();
  float* dst = partials + (size_t) blockIdx.x * M * V;
  for (int i = threadIdx.x; i < M * V; i += 256) dst[i] = s_acc[i];
}
__global__ void __launch_bounds__(256) dist_weights_finalize_kernel(int M, int dim, int nblk, const float* __restrict__ partials,
    float* __restrict__ g_joints, float* __restrict__ g_radius, float* __restrict__ g_kweight,
    const float* __restrict__ radius, const float* __restrict__ kweight, int activate, int accumulate_joints) {
  const int V = dim + 2, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * V) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += partials[(size_t) b * M * V + i];
  const int j = i / V, c = i % V;
  if (c < dim) {
    if (g_joints) g_joints[(size_t) j * dim + c] = accumulate_joints ? g_joints[(size_t) j * dim + c] + s : s;
  } else if (c == dim) {
    // activate: the gradient w.r.t. the RAW parameter: d exp(x) = exp(x), d sigmoid(x) = s (1 - s)
    if (g_radius) g_radius[j] = (activate && radius) ? s * expf(radius[j]) : s;
  } else if (g_kweight) {
    float d = 1.f;
    if (activate && kweight) {
      const float sg = 1.0f / (1.0f + expf(-kweight[j]));
      d = sg * (1.f - sg);
    }
    g_kweight[j] = s * d;
  }
}

// K nearest bones + LBS weights in one pass (the two calls of calc_LBS_weight, sk_gs.py:757,769-770): top-K as
// knn_bones_kernel (dim = 3), then softmax of the gathered logits; indices and weights leave through LDS so that a
// workgroup stores contiguous spans instead of K strided 8-byte pieces per lane.
template <int KCAP>
__global__ void __launch_bounds__(256) knn_weights_kernel(int P, int M, int K, const float* __restrict__ points,
    const float* __restrict__ joints, const float* __restrict__ sp_W, int64_t* __restrict__ out_idx,
    float* __restrict__ out_weights) {
  extern __shared__ float s_dyn[];
  float* s_j       = s_dyn;                                                  // [M][3]
  float* s_w       = s_dyn + ((M * 3 + 3) & ~3);                             // [256][K]
  int64_t* s_idx   = reinterpret_cast<int64_t*>(s_w + ((256 * K + 3) & ~3));  // [256][K]
  for (int i = threadIdx.x; i < M * 3; i += 256) s_j[i] = joints[i];
  __syncthreads();
  const int p0 = blockIdx.x * 256, n = p0 + threadIdx.x;
  if (n < P) {
    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int k = 0; k < KCAP; ++k) bd[k] = __builtin_inff(), bi[k] = 0;
    const float p0x = points[3 * (size_t) n], p1 = points[3 * (size_t) n + 1], p2 = points[3 * (size_t) n + 2];
    for (int j = 0; j < M; ++j) {
      const float d0 = p0x - s_j[3 * j], d1 = p1 - s_j[3 * j + 1], d2 = p2 - s_j[3 * j + 2];
      float d = 0.f;
      d += d0 * d0;
      d += d1 * d1;
      d += d2 * d2;
      float cd = d;
      int ci   = j;
      topk_insert<KCAP>(bd, bi, cd, ci);
    }
    float l[KCAP];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      l[k] = k < K ? sp_W[(size_t) n * M + bi[k]] : -INFINITY;
      mx   = fmaxf(mx, l[k]);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KCAP; ++k) {
      l[k] = k < K ? expf(l[k] - mx) : 0.f;
      sum += l[k];
    }
#pragma unroll
    for (int k = 0; k < KCAP; ++k)
      if (k < K) s_w[threadIdx.x * K + k] = l[k] / sum, s_idx[threadIdx.x * K + k] = bi[k];
  }
  __syncthreads();
  const int cnt = min(256, P - p0) * K;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    out_weights[(size_t) p0 * K + i] = s_w[i];
    out_idx[(size_t) p0 * K + i]     = s_idx[i];
  }
}


// knn_weights_kernel + deform_forward_kernel<true> in one launch (the sk stage runs them back to back on the same
// Gaussians, sk_gs.py:757-770 then :1143-1150): the K (index, weight) pairs stay in registers between the two halves, so
// the weights / indices are written for the backward but never re-read, and one launch disappears.  Same arithmetic, in
// the same order, as the two kernels (the tests compare the three entry points bit for bit).
template <int KCAP>
__global__ void __launch_bounds__(256) knn_deform_forward_kernel(int P, int M, int K, const float* __restrict__ points,
    const float* __restrict__ joints, const float* __restrict__ sp_W, const float* __restrict__ bone_T,
    const float* __restrict__ bone_drot, const float* __restrict__ bone_dscale, const float* __restrict__ xyz,
    const float* __restrict__ log_scale, const float* __restrict__ rot, const float* __restrict__ opacity_logit,
    int64_t* __restrict__ out_idx, float* __restrict__ out_weights, float* __restrict__ means, float* __restrict__ scales,
    float* __restrict__ rotations, float* __restrict__ opacity, const int32_t* __restrict__ live) {
  if (live) P = min(P, live[0]);  // the number of Gaussians is a device word: one captured graph survives densification
  if ((int) (blockIdx.x * 256) >= P) return;  // slack rows of the capacity
  extern __shared__ float s_dyn[];
  float* s_j       = s_dyn;                                                  // [M][3]
  float* s_bones   = s_dyn + ((M * 3 + 3) & ~3);                             // [M][14]
  float* s_w       = s_bones + ((M * BONE_F + 3) & ~3);                      // [256][K]
  int64_t* s_idx   = reinterpret_cast<int64_t*>(s_w + ((256 * K + 3) & ~3));  // [256][K]
  for (int i = threadIdx.x; i < M * 3; i += 256) s_j[i] = joints[i];
  for (int j = threadIdx.x; j < M; j += 256) load_bone(bone_T, bone_drot, bone_dscale, j, s_bones + j * BONE_F);
  __syncthreads();
  const int p0 = blockIdx.x * 256, n = p0 + threadIdx.x;
  if (n < P) {
    const float p[3] = {points[3 * (size_t) n], points[3 * (size_t) n + 1], points[3 * (size_t) n + 2]};
    float w[KCAP], sx[3], sr[4], ss[3];
    int bi[KCAP];
    knn_softmax_skin_lane<KCAP>(M, K, s_j, s_bones, p, [&](int j) { return sp_W[(size_t) n * M + j]; }, w, bi, sx, sr, ss);
#pragma unroll
    for (int k = 0; k < KCAP; ++k)
      if (k < K) s_w[threadIdx.x * K + k] = w[k], s_idx[threadIdx.x * K + k] = bi[k];
    const float x3[3] = {xyz[3 * n], xyz[3 * n + 1], xyz[3 * n + 2]};
    const float ls[3] = {log_scale[3 * n], log_scale[3 * n + 1], log_scale[3 * n + 2]};
    float mo[3], so[3], oo;
    float4 ro;
    deform_activate_lane(p, sx, sr, ss, x3, ls, reinterpret_cast<const float4*>(rot)[n], opacity_logit[n], mo, so, ro, oo);
#pragma unroll
    for (int c = 0; c < 3; ++c) means[3 * n + c] = mo[c], scales[3 * n + c] = so[c];
    reinterpret_cast<float4*>(rotations)[n] = ro;
    opacity[n]                              = oo;
  }
  __syncthreads();
  const int cnt = min(256, P - p0) * K;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    out_weights[(size_t) p0 * K + i] = s_w[i];
    out_idx[(size_t) p0 * K + i]     = s_idx[i];
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
        means, scales, rotations, opacity, d_xyz, d_rot, d_scale, in.largest ? 1 : 0);
  else
    hipLaunchKernelGGL(deform_forward_kernel<false>, grid, block, 0, s, in.P, in.K, in.M, in.points, in.weights,
        in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.xyz, in.log_scale, in.rot, in.opacity_logit, means, scales,
        rotations, opacity, d_xyz, d_rot, d_scale, in.largest ? 1 : 0);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

constexpr int WIDE_MAX_BLOCKS = 512;
static int wide_blocks(int P) { return std::max(1, std::min((P + DEFORM_THREADS - 1) / DEFORM_THREADS, WIDE_MAX_BLOCKS)); }
size_t deform_backward_workspace_bytes(int P, int M) {
  if (M > MOM_MAX_BONES) return M <= MAX_LDS_BONES / 2 ? align256((size_t) wide_blocks(P) * M * BONE_F * 4) + 256 : 256;
  const size_t nblk = (size_t) (P + DEFORM_THREADS - 1) / DEFORM_THREADS;
  return align256(nblk * (size_t) M * MOM_F * 4) + 256;
}

int launch_deform_backward(const skgs_deform_inputs& in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, void* workspace,
    hipStream_t s, float* g_sp_W, float* g_logits) {
  if ((g_sp_W || g_logits) && (in.M > MOM_MAX_BONES || in.K > PREF_K))
    return set_error("deform backward with the logit gradient folded in needs M <= %d and K <= %d (got %d, %d)",
        MOM_MAX_BONES, PREF_K, in.M, in.K);
  if (in.P == 0) {  // outputs are always written completely
    if (fill_u32(g_bone_T, 0u, (size_t) in.M * 7, s) || fill_u32(g_bone_drot, 0u, (size_t) in.M * 4, s) ||
        fill_u32(g_bone_dscale, 0u, (size_t) in.M * 3, s))
      return 1;
    return 0;
  }
  dim3 grid((in.P + DEFORM_THREADS - 1) / DEFORM_THREADS), block(DEFORM_THREADS);
  if (in.M <= MOM_MAX_BONES) {
    const size_t lds = deform_bwd_lds_bytes(in.M);
    float* partials  = reinterpret_cast<float*>(workspace);
    DeformBwdArgs a{in.K, in.M, in.points, in.weights, in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.log_scale, in.rot,
        in.opacity_logit, g_weights, g_xyz, g_log_scale, g_rot, g_opacity_logit, partials, g_sp_W, g_logits};
    {
      ProfScope prof(K_DEFORM_BWD, s);
      hipLaunchKernelGGL(deform_backward_moments_kernel, grid, block, lds, s, in.P, a, g_means, g_scales, g_rotations, g_opacity,
          in.live_count);
    }
    SKGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(deform_backward_finalize_kernel, dim3(in.M), dim3(256), 0, s, in.M, (int) grid.x, partials, in.bone_T,
        g_bone_T, g_bone_drot, g_bone_dscale);
    SKGS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // many bones (superpoint stage)
  if (in.M <= MAX_LDS_BONES / 2) {  // LDS tables per workgroup -> partials -> fixed-order sum (no memset, no global atomics)
    const size_t table = (size_t) in.M * BONE_F * 4;
    int ncopy = 1;
    while (ncopy < 16 && table * (1 + 2 * ncopy) <= 56 * 1024) ncopy *= 2;
    const int nblk  = wide_blocks(in.P);
    float* partials = reinterpret_cast<float*>(workspace);
    {
      ProfScope prof(K_DEFORM_BWD, s);
      hipLaunchKernelGGL(deform_backward_kernel<true>, dim3(nblk), block, table * (1 + ncopy), s, in.P, in.K, in.M, in.points,
          in.weights, in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.log_scale, in.rot, in.opacity_logit, g_means,
          g_scales, g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot, g_bone_dscale, g_xyz, g_log_scale, g_rot,
          g_opacity_logit, ncopy, partials);
    }
    SKGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(deform_backward_wide_finalize_kernel, dim3((in.M * BONE_F + 255) / 256), dim3(256), 0, s, in.M, nblk,
        partials, g_bone_T, g_bone_drot, g_bone_dscale);
    SKGS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // more bones than the LDS tables hold: scatter with global atomics into zeroed outputs
  if (fill_u32(g_bone_T, 0u, (size_t) in.M * 7, s) || fill_u32(g_bone_drot, 0u, (size_t) in.M * 4, s) ||
      fill_u32(g_bone_dscale, 0u, (size_t) in.M * 3, s))
    return 1;
  ProfScope prof(K_DEFORM_BWD, s);
  hipLaunchKernelGGL(deform_backward_kernel<false>, grid, block, 0, s, in.P, in.K, in.M, in.points, in.weights,
      in.indices, in.bone_T, in.bone_drot, in.bone_dscale, in.log_scale, in.rot, in.opacity_logit, g_means, g_scales,
      g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot, g_bone_dscale, g_xyz, g_log_scale, g_rot,
      g_opacity_logit, 1, (float*) nullptr);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

// the second launch of the moments path alone: the first ran as a job of the rasterizer's per-Gaussian backward launch
// (preprocess.hip, skgs_raster_grads.deform_backward_job), grid = ceil(P / DEFORM_BWD_THREADS) workgroups
int launch_deform_backward_finalize(const skgs_deform_inputs& in, void* workspace, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, hipStream_t s) {
  const int nblk = (in.P + DEFORM_BWD_THREADS - 1) / DEFORM_BWD_THREADS;
  hipLaunchKernelGGL(deform_backward_finalize_kernel, dim3(in.M), dim3(256), 0, s, in.M, nblk, reinterpret_cast<float*>(workspace),
      in.bone_T, g_bone_T, g_bone_drot, g_bone_dscale);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
int deform_backward_job_max_bones() { return MOM_MAX_BONES; }
int deform_backward_job_max_k() { return PREF_K; }

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

int launch_lbs_weights_forward(int P, int M, int K, const float* sp_W, const int64_t* indices, float* weights, hipStream_t s) {
  if (P == 0) return 0;
  if (K > KNN_MAXK || K < 1) return set_error("lbs_weights: K must be in [1,%d] (got %d)", KNN_MAXK, K);
  hipLaunchKernelGGL(lbs_weights_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, M, K, sp_W, indices, weights);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_lbs_weights_backward(int P, int M, int K, const float* weights, const int64_t* indices, const float* g_weights,
    float* g_sp_W, hipStream_t s) {
  if (P == 0 || M == 0) return 0;
  if (M > SKGS_FUSED_LBS_MAX_BONES) {  // superpoint-sized M: no LDS row staging
    const long long n = (long long) P * ((M + 3) / 4);
    hipLaunchKernelGGL(lbs_logits_dense_wide_kernel<true>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, P, M, K, weights,
        indices, g_weights, g_sp_W);
    SKGS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(lbs_weights_backward_kernel, dim3((P + 255) / 256), dim3(256), (size_t) M * 256 * 4, s, P, M, K, weights,
      indices, g_weights, g_sp_W);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_knn_lbs_weights(int P, int M, int K, const float* points, const float* joints, const float* sp_W, int64_t* out_idx,
    float* out_weights, hipStream_t s) {
  if (P == 0) return 0;
  if (K > KNN_MAXK || K < 1 || K > M) return set_error("knn_lbs_weights: K must be in [1,min(%d,M)] (got %d)", KNN_MAXK, K);
  const size_t lds = ((size_t) ((M * 3 + 3) & ~3) + ((256 * K + 3) & ~3)) * 4 + (size_t) 256 * K * 8;
  if (lds > 60 * 1024) return set_error("knn_lbs_weights: M = %d too large for the LDS joint table", M);
  ProfScope prof(K_KNN, s);
#define SKGS_KNNW(KCAP_)                                                                                            \
  hipLaunchKernelGGL(knn_weights_kernel<KCAP_>, dim3((P + 255) / 256), dim3(256), lds, s, P, M, K, points, joints, sp_W, \
      out_idx, out_weights)
  if (K <= 4)
    SKGS_KNNW(4);
  else if (K <= 5)
    SKGS_KNNW(5);
  else if (K <= 8)
    SKGS_KNNW(8);
  else
    SKGS_KNNW(KNN_MAXK);
#undef SKGS_KNNW
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_knn_dist_weights_forward(int P, int M, int K, int dim, const float* points, const float* joints, const float* radius,
    const float* kweight, float temperature, int activate, int64_t* out_idx, float* out_weights, float* out_dist, hipStream_t s) {
  if (P == 0) return 0;
  if (K > KNN_MAXK || K < 1 || K > M) return set_error("knn_dist_weights: K must be in [1,min(%d,M)] (got %d)", KNN_MAXK, K);
  if (dim < 1 || dim > 16) return set_error("knn_dist_weights: dim must be in [1,16] (got %d)", dim);
  const size_t lds = (size_t) M * dim * 4;
  const int use_lds = lds <= 48 * 1024;
  ProfScope prof(K_KNN, s);
#define SKGS_KNND(KCAP_)                                                                                                     \
  hipLaunchKernelGGL(knn_dist_weights_kernel<KCAP_>, dim3((P + 255) / 256), dim3(256), use_lds ? lds : 0, s, P, M, K, dim, points, \
      joints, radius, kweight, temperature, out_idx, out_weights, out_dist, use_lds, activate)
  if (K <= 4)
    SKGS_KNND(4);
  else if (K <= 5)
    SKGS_KNND(5);
  else if (K <= 8)
    SKGS_KNND(8);
  else
    SKGS_KNND(KNN_MAXK);
#undef SKGS_KNND
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

static int dist_weights_blocks(int P) { return std::max(1, std::min((P + 255) / 256, DW_MAX_BLOCKS)); }
size_t knn_dist_weights_workspace_bytes(int P, int M, int dim) {
  return (size_t) dist_weights_blocks(P) * M * (dim + 2) * sizeof(float);
}
int launch_knn_dist_weights_backward(int P, int M, int K, int dim, const float* points, const float* joints, const float* radius,
    const float* kweight, float temperature, int activate, int accumulate_joints, const float* weights, const int64_t* indices,
    const float* nn_dist, const float* g_weights, float* g_points, float* g_joints, float* g_radius, float* g_kweight,
    void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (K > KNN_MAXK || K < 1) return set_error("knn_dist_weights_backward: K must be in [1,%d] (got %d)", KNN_MAXK, K);
  if (dim < 1 || dim > 16) return set_error("knn_dist_weights_backward: dim must be in [1,16] (got %d)", dim);
  const int V = dim + 2, nblk = dist_weights_blocks(P);
  const size_t lds = (size_t) M * V * 4;
  if (lds > 60 * 1024) return set_error("knn_dist_weights_backward: %d bones x %d values do not fit the LDS accumulators", M, V);
  if (workspace_bytes < knn_dist_weights_workspace_bytes(P, M, dim) || !workspace)
    return set_error("knn_dist_weights_backward: workspace too small (%zu bytes)", workspace_bytes);
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(dist_weights_backward_kernel, dim3(nblk), dim3(256), lds, s, P, M, K, dim, points, joints, radius, kweight,
      temperature, weights, indices, nn_dist, g_weights, g_points, partials, activate);
  hipLaunchKernelGGL(dist_weights_finalize_kernel, dim3((M * V + 255) / 256), dim3(256), 0, s, M, dim, nblk, partials, g_joints,
      g_radius, g_kweight, radius, kweight, activate, accumulate_joints);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_knn_deform_forward(int P, int M, int K, const float* points, const float* joints, const float* sp_W,
    const float* bone_T, const float* bone_drot, const float* bone_dscale, const float* xyz, const float* log_scale,
    const float* rot, const float* opacity_logit, int64_t* out_idx, float* out_weights, float* means, float* scales,
    float* rotations, float* opacity, const int32_t* live_count, hipStream_t s) {
  if (P == 0) return 0;
  if (K > KNN_MAXK || K < 1 || K > M) return set_error("knn_deform_forward: K must be in [1,min(%d,M)] (got %d)", KNN_MAXK, K);
  const size_t lds = ((size_t) ((M * 3 + 3) & ~3) + ((M * BONE_F + 3) & ~3) + ((256 * K + 3) & ~3)) * 4 + (size_t) 256 * K * 8;
  if (lds > 60 * 1024) return set_error("knn_deform_forward: M = %d too large for the LDS tables", M);
  ProfScope prof(K_DEFORM_FWD, s);
#define SKGS_KNND(KCAP_)                                                                                                 \
  hipLaunchKernelGGL(knn_deform_forward_kernel<KCAP_>, dim3((P + 255) / 256), dim3(256), lds, s, P, M, K, points, joints, \
      sp_W, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot, opacity_logit, out_idx, out_weights, means, scales,      \
      rotations, opacity, live_count)
  if (K <= 4)
    SKGS_KNND(4);
  else if (K <= 5)
    SKGS_KNND(5);
  else if (K <= 8)
    SKGS_KNND(8);
  else
    SKGS_KNND(KNN_MAXK);
#undef SKGS_KNND
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_lbs_weights_backward_compact(int P, int K, const float* weights, const float* g_weights, float* g_logits,
    hipStream_t s) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(lbs_weights_backward_compact_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, K, weights, g_weights,
      g_logits);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_lbs_logits_scatter(int P, int M, int K, const int64_t* indices, const float* g_logits, float* g_sp_W,
    hipStream_t s) {
  if (P == 0 || M == 0) return 0;
  if (M > SKGS_FUSED_LBS_MAX_BONES) {
    const long long n = (long long) P * ((M + 3) / 4);
    hipLaunchKernelGGL(lbs_logits_dense_wide_kernel<false>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, P, M, K, nullptr,
        indices, g_logits, g_sp_W);
    SKGS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(lbs_logits_scatter_kernel, dim3((P + 255) / 256), dim3(256), (size_t) M * 256 * 4, s, P, M, K, indices,
      g_logits, g_sp_W);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
