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

// ------------------------------------------------------------------------------ deform backward: bone gradients by moments
// (the algorithm: deform.hip, above deform_backward_moments_kernel.)  The body is shared by that kernel and by the rasterizer's
// per-Gaussian backward launch (preprocess.hip), which runs it on the gradients it has just produced, from registers.
constexpr int PREF_K             = 8;  // neighbour slots prefetched into registers (K is 5 in every shipped config)
constexpr int MOM_F              = 19;
constexpr int MOM_U              = 20;  // padded row of U
constexpr int MOM_MAX_BONES      = 64;
constexpr int DEFORM_BWD_THREADS = 256;  // 4 waves: the LDS carving below and the partial sums assume it

struct DeformBwdArgs {
  int K, M;
  const float *points, *weights;
  const int64_t* indices;
  const float *bone_T, *bone_drot, *bone_dscale, *log_scale, *rot, *opacity_logit;
  float *g_weights, *g_xyz, *g_log_scale, *g_rot, *g_opacity_logit;
  float* partials;  // [gridDim.x][M][19]
  float* g_sp_W;    // [P,M] or NULL
  float* g_logits;  // [P,K] or NULL; both need K <= PREF_K
};
// what a lane requests up front (one round trip instead of a chain of dependent ones: these launches run at ~1.5 waves per
// SIMD, their duration is the length of a lane's dependency chain).  The first PREF_K neighbour slots live in registers; a K
// beyond that reads the rest in place.
struct DeformBwdLane {
  float p[3], ls[3], ol;
  float4 r4;
  int j[PREF_K];
  float w[PREF_K];
};
inline size_t deform_bwd_lds_bytes(int M) {
  const int Mp = (M + 3) & ~3;
  return ((size_t) M * BONE_F + 4 * 64 * (size_t) Mp + 4 * 64 * MOM_U + 4 * (size_t) M * MOM_F) * 4;
}
__device__ __forceinline__ void deform_bwd_prefetch(const DeformBwdArgs& a, int n, bool valid, DeformBwdLane& L) {
#pragma unroll
  for (int c = 0; c < 3; ++c) L.p[c] = 0.f, L.ls[c] = 0.f;
  L.ol = 0.f, L.r4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < PREF_K; ++q) L.j[q] = 0, L.w[q] = 0.f;
  if (valid) {
#pragma unroll
    for (int q = 0; q < PREF_K; ++q)
      if (q < a.K) L.j[q] = (int) a.indices[(size_t) n * a.K + q], L.w[q] = a.weights[(size_t) n * a.K + q];
#pragma unroll
    for (int c = 0; c < 3; ++c) L.p[c] = a.points[3 * n + c], L.ls[c] = a.log_scale[3 * n + c];
    L.r4 = reinterpret_cast<const float4*>(a.rot)[n], L.ol = a.opacity_logit[n];
  }
}
// a workgroup of the capacity's slack rows: its partial is zero
__device__ __forceinline__ void deform_bwd_zero_partials(const DeformBwdArgs& a) {
  for (int o = threadIdx.x; o < a.M * MOM_F; o += DEFORM_BWD_THREADS) a.partials[(size_t) blockIdx.x * a.M * MOM_F + o] = 0.f;
}
// The whole workgroup (DEFORM_BWD_THREADS threads, Gaussian n = blockIdx.x * DEFORM_BWD_THREADS + threadIdx.x, P the live
// count) with the upstream gradients of its Gaussian in registers; s_mem: deform_bwd_lds_bytes(M) of LDS nobody else uses
// from here on.  Contains workgroup barriers.
__device__ __forceinline__ void deform_bwd_moments(const DeformBwdArgs& a, int P, float* s_mem, const DeformBwdLane& L,
    const float (&g_dx_in)[3], const float (&g_ds_in)[3], float4 gr4_in, float go_in) {
  const int Mp    = (a.M + 3) & ~3;                 // weight rows padded to float4
  float* s_bones  = s_mem;                        // [a.M][14]
  float* s_w      = s_bones + a.M * BONE_F;         // [4 waves][64][Mp]
  float* s_u      = s_w + 4 * 64 * Mp;            // [4 waves][64][20]
  float* s_part   = s_u + 4 * 64 * MOM_U;         // [4 waves][a.M*19]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* my_w = s_w + (size_t) (wave * 64 + lane) * Mp;
  float* my_u = s_u + (size_t) (wave * 64 + lane) * MOM_U;
  const int n = blockIdx.x * DEFORM_BWD_THREADS + threadIdx.x;
  for (int j = threadIdx.x; j < a.M; j += DEFORM_BWD_THREADS) load_bone(a.bone_T, a.bone_drot, a.bone_dscale, j, s_bones + j * BONE_F);
  for (int i = 0; i < Mp; i += 4) *reinterpret_cast<float4*>(my_w + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const bool want_logits = a.g_sp_W != nullptr || a.g_logits != nullptr;
  float lw[PREF_K], lg[PREF_K];
  int lj[PREF_K];
#pragma unroll
  for (int q = 0; q < PREF_K; ++q) lw[q] = 0.f, lg[q] = 0.f, lj[q] = 0;
  float u[MOM_U];
#pragma unroll
  for (int c = 0; c < MOM_U; ++c) u[c] = 0.f;
  if (n < P) {
    const float p[3] = {L.p[0], L.p[1], L.p[2]};
    float sr[4] = {0, 0, 0, 0};
    auto blend_rot = [&](int j, float w) {
      const float* b = s_bones + j * BONE_F;
      sr[0] += b[7] * w, sr[1] += b[8] * w, sr[2] += b[9] * w, sr[3] += b[10] * w;
      my_w[j] += w;  // own row: plain read-modify-write (KNN ids are distinct, += keeps it right if they are not)
    };
#pragma unroll
    for (int q = 0; q < PREF_K; ++q)
      if (q < a.K) blend_rot(L.j[q], L.w[q]);
    for (int k = PREF_K; k < a.K; ++k) blend_rot((int) a.indices[(size_t) n * a.K + k], a.weights[(size_t) n * a.K + k]);
    const float4 r4 = L.r4, gr4 = gr4_in;
    const float v[4] = {r4.x + sr[0], r4.y + sr[1], r4.z + sr[2], r4.w + sr[3]};
    const float gr[4] = {gr4.x, gr4.y, gr4.z, gr4.w};
    const float nv    = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    float g_v[4];
    if (nv > 1e-12f) {
      const float uq[4] = {v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv};
      const float dot   = uq[0] * gr[0] + uq[1] * gr[1] + uq[2] * gr[2] + uq[3] * gr[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) g_v[c] = (gr[c] - uq[c] * dot) / nv;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) g_v[c] = gr[c] / 1e-12f;
    }
    const float g_dx[3] = {g_dx_in[0], g_dx_in[1], g_dx_in[2]};
    const float g_ds[3] = {g_ds_in[0], g_ds_in[1], g_ds_in[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.g_xyz[3 * n + c]       = g_dx[c];
      a.g_log_scale[3 * n + c] = g_ds[c] * expf(L.ls[c]);
    }
    reinterpret_cast<float4*>(a.g_rot)[n] = make_float4(g_v[0], g_v[1], g_v[2], g_v[3]);
    const float sg     = 1.0f / (1.0f + expf(-L.ol));
    a.g_opacity_logit[n] = go_in * sg * (1.0f - sg);
    float dot = 0.f;
    // dL/dw[p,k] = g_dx . (T_j p) + g_v . d_rot_j + g_ds . d_scale_j
    auto weight_grad = [&](int j) {
      const float* b = s_bones + j * BONE_F;
      float y[3];
      se3_act(b, p, y);
      float gw = g_dx[0] * y[0] + g_dx[1] * y[1] + g_dx[2] * y[2];
#pragma unroll
      for (int c = 0; c < 4; ++c) gw += g_v[c] * b[7 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) gw += g_ds[c] * b[11 + c];
      return gw;
    };
#pragma unroll
    for (int q = 0; q < PREF_K; ++q) {
      if (q < a.K) {
        const float gw = weight_grad(L.j[q]);
        if (a.g_weights) a.g_weights[(size_t) n * a.K + q] = gw;
        // softmax backward of the sp_W branch (lbs_weights_backward_kernel): same order of operations
        lw[q] = L.w[q], lg[q] = gw, lj[q] = L.j[q];
        dot += L.w[q] * gw;
      }
    }
    for (int k = PREF_K; k < a.K; ++k) {  // (the logit gradient needs a.K <= PREF_K: enforced by the launcher)
      const float gw = weight_grad((int) a.indices[(size_t) n * a.K + k]);
      if (a.g_weights) a.g_weights[(size_t) n * a.K + k] = gw;
    }
    if (want_logits) {
#pragma unroll
      for (int q = 0; q < PREF_K; ++q) lg[q] = q < a.K ? lw[q] * (lg[q] - dot) : 0.f;
      if (a.g_logits)
#pragma unroll
        for (int q = 0; q < PREF_K; ++q)
          if (q < a.K) a.g_logits[(size_t) n * a.K + q] = lg[q];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[a] = g_dx[a];
#pragma unroll
      for (int c = 0; c < 3; ++c) u[3 + 3 * a + c] = g_dx[a] * p[c];
      u[16 + a] = g_ds[a];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) u[12 + c] = g_v[c];
  }
#pragma unroll
  for (int c = 0; c < MOM_U; c += 4) *reinterpret_cast<float4*>(my_u + c) = make_float4(u[c], u[c + 1], u[c + 2], u[c + 3]);
  __syncthreads();  // (only the own wave's rows are read below; the barrier also orders the LDS traffic)
  // ---- Mom_wave[bone][component] = sum over the wave's 64 Gaussians of w[q][bone] u[q][component]: outer products on the
  // matrix cores.  v_mfma_f32_4x4x1 holds 16 independent 4 x 4 blocks = (4 bones) x (4 components) each; one Gaussian per
  // instruction; lane 4 b + i feeds bone 4 bg + i as A and component 4 cg + i as B of block b and receives row i' of the
  // block in VGPR i'.  (On the VALU every lane owned ~6 outputs and read 2 x 64 LDS words for each: 768 ds_read_b32 and 384
  // FMAs per lane; now 256 reads and 128 MFMAs for a.M = 20.)
  const int n_out   = a.M * MOM_F;
  const float* w0   = s_w + (size_t) wave * 64 * Mp;
  const float* u0   = s_u + (size_t) wave * 64 * MOM_U;
  float* part       = s_part + (size_t) wave * n_out;
  {
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int NCG = MOM_U / 4;
    const int nblk = (Mp / 4) * NCG, li = lane & 3, lb = lane >> 2;
    for (int r0 = 0; r0 < nblk; r0 += 16) {
      const int blk   = r0 + lb;
      const bool live = blk < nblk;
      const int bg = live ? blk / NCG : 0, cg = live ? blk - (blk / NCG) * NCG : 0;
      const float* wa = w0 + 4 * bg + li;
      const float* ub = u0 + 4 * cg + li;
      f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int q = 0; q < 64; q += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[q * Mp], ub[q * MOM_U], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[(q + 1) * Mp], ub[(q + 1) * MOM_U], acc1, 0, 0, 0);
      }
      if (live) {
        const int comp = 4 * cg + li;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int bone = 4 * bg + i;
          if (bone < a.M && comp < MOM_F) part[bone * MOM_F + comp] = acc0[i] + acc1[i];
        }
      }
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += DEFORM_BWD_THREADS)
    a.partials[(size_t) blockIdx.x * n_out + o] = (s_part[o] + s_part[n_out + o]) + (s_part[2 * n_out + o] + s_part[3 * n_out + o]);
  // ---- dense logit-gradient rows (lbs_weights_backward_kernel folded in): the weight rows in LDS are no longer needed,
  // each lane rebuilds its row there as the gradient row, the workgroup stores its 256 rows as one contiguous span
  if (a.g_sp_W) {
    for (int i = 0; i < Mp; i += 4) *reinterpret_cast<float4*>(my_w + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < P) {
#pragma unroll
      for (int q = 0; q < PREF_K; ++q)
        if (q < a.K) my_w[lj[q]] += lg[q];  // KNN ids are distinct; += keeps the gather-backward semantics if they are not
    }
    __syncthreads();
    const int p0   = blockIdx.x * DEFORM_BWD_THREADS;
    const int rows = min(DEFORM_BWD_THREADS, P - p0);
    float* dst     = a.g_sp_W + (size_t) p0 * a.M;
    for (int i = threadIdx.x; i < rows * a.M; i += DEFORM_BWD_THREADS) stream_store<NT_DEFORM_BWD>(dst + i, s_w[(i / a.M) * Mp + (i % a.M)]);
  }
}

// ------------------------------------------------------------------------------ superpoint stage: the rows pass of the backward
// (sp_backward.hip: one lane per Gaussian -- its four parameter gradients, g_weights, the weighting's chain rule,
// hyper_feature.grad and the compact payload U / V of the bones pass).  Shared by sp_backward_rows_kernel and by the rasterizer's
// per-Gaussian backward launch (preprocess.hip), which runs it on the gradients it has just produced.
constexpr int SP_MAXK = 16;
constexpr int SP_MAXF = 8;
constexpr int SP_UROW = 12;  // per-Gaussian payload row: g_dx 3 | g_v 4 | g_ds 3 | pad 2
// warp_method `largest` (sk_gs.py:811-816,849-850): the slot of a row's largest weight, the first of equal ones (torch.argmax)
__device__ __forceinline__ int argmax_slot(const float* __restrict__ w, int K) {
  int best = 0;
  float wb = w[0];
  for (int k = 1; k < K; ++k) {
    const float wk = w[k];
    if (wk > wb) wb = wk, best = k;
  }
  return best;
}
struct SpRowsArgs {
  int K, M, largest;
  const float *points, *weights;
  const int64_t* indices;
  const float *nn_dist, *bone_T, *bone_drot, *bone_dscale, *log_scale, *rot, *opacity_logit, *feature, *sp_feature, *radius_raw,
      *kweight_raw;
  float temperature;
  int logits;
  float *g_weights, *g_xyz, *g_log_scale, *g_rot, *g_opacity_logit, *g_feature, *U, *V;
  const float* g_weights_extra;
};
// the arguments from the public job (U and V are the first two pieces of its workspace: sp_backward.hip carves the same way)
inline SpRowsArgs sp_rows_args(const skgs_sp_skinning_job& j) {
  const skgs_deform_inputs* in = j.in;
  const size_t P = (size_t) (in->P > 1 ? in->P : 1);
  char* wsp      = reinterpret_cast<char*>(j.workspace);
  float* U       = reinterpret_cast<float*>(wsp);
  float* V       = reinterpret_cast<float*>(wsp + align256(P * SP_UROW * 4));
  return SpRowsArgs{in->K, in->M, (int) in->largest, in->points, in->weights, in->indices, j.nn_dist, in->bone_T, in->bone_drot, in->bone_dscale,
      in->log_scale, in->rot, in->opacity_logit, j.feature, j.sp_feature, j.sp_radius_raw, j.sp_weight_raw, j.temperature,
      (int) j.logit_weighting, j.g_weights, j.g_xyz, j.g_log_scale, j.g_rot, j.g_opacity_logit, j.g_feature, U, V, j.g_weights_extra};
}
inline size_t sp_rows_lds_bytes(int M) { return (size_t) M * BONE_F * 4; }
// s_bones: [M][BONE_F] staged by the caller (load_bone), n: the lane's Gaussian (< P)
template <int F>
__device__ __forceinline__ void sp_rows_lane(const SpRowsArgs& ja, const float* s_bones, int n, const float (&g_dx_in)[3],
    const float (&g_ds_in)[3], float4 gr4_in, float go_in) {
  const float p[3] = {ja.points[3 * n], ja.points[3 * n + 1], ja.points[3 * n + 2]};
  int jj[SP_MAXK];
  float ww[SP_MAXK];
#pragma unroll
  for (int k = 0; k < SP_MAXK; ++k) {
    jj[k] = k < ja.K ? (int) ja.indices[(size_t) n * ja.K + k] : 0;
    ww[k] = k < ja.K ? ja.weights[(size_t) n * ja.K + k] : 0.f;
  }
  // ---- the Gaussian's own gradients (deform.hip::deform_backward_kernel, same expressions)
  float sr[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < SP_MAXK; ++k)
    if (k < ja.K) {
      const float* b = s_bones + jj[k] * BONE_F;
      sr[0] += b[7] * ww[k], sr[1] += b[8] * ww[k], sr[2] += b[9] * ww[k], sr[3] += b[10] * ww[k];
    }
  const float4 r4  = reinterpret_cast<const float4*>(ja.rot)[n];
  const float4 gr4 = gr4_in;
  const float v[4]  = {r4.x + sr[0], r4.y + sr[1], r4.z + sr[2], r4.w + sr[3]};
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
  const float g_dx[3] = {g_dx_in[0], g_dx_in[1], g_dx_in[2]};
  const float g_ds[3] = {g_ds_in[0], g_ds_in[1], g_ds_in[2]};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    ja.g_xyz[3 * n + c]       = g_dx[c];
    ja.g_log_scale[3 * n + c] = g_ds[c] * expf(ja.log_scale[3 * n + c]);
  }
  reinterpret_cast<float4*>(ja.g_rot)[n] = make_float4(g_v[0], g_v[1], g_v[2], g_v[3]);
  const float sg     = 1.0f / (1.0f + expf(-ja.opacity_logit[n]));
  ja.g_opacity_logit[n] = go_in * sg * (1.0f - sg);
  float* un = ja.U + (size_t) n * SP_UROW;
  reinterpret_cast<float4*>(un)[0] = make_float4(g_dx[0], g_dx[1], g_dx[2], g_v[0]);
  reinterpret_cast<float4*>(un)[1] = make_float4(g_v[1], g_v[2], g_v[3], g_ds[0]);
  reinterpret_cast<float4*>(un)[2] = make_float4(g_ds[1], g_ds[2], 0.f, 0.f);
  // ---- g_weights[k] = g_dx . (T_j p) + g_v . d_rot_j + g_ds . d_scale_j
  float gw[SP_MAXK];
#pragma unroll
  for (int k = 0; k < SP_MAXK; ++k) {
    gw[k] = 0.f;
    if (k < ja.K) {
      const float* b = s_bones + jj[k] * BONE_F;
      float y[3];
      se3_act(b, p, y);
      float a = g_dx[0] * y[0] + g_dx[1] * y[1] + g_dx[2] * y[2];
      if (ja.largest) a = 0.f;  // the position follows ONE bone: the weights reach the loss through the rotation / scale blend only
#pragma unroll
      for (int c = 0; c < 4; ++c) a += g_v[c] * b[7 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) a += g_ds[c] * b[11 + c];
      if (ja.g_weights_extra) a += ja.g_weights_extra[(size_t) n * ja.K + k];  // (a cotangent on the weights from outside the skinning)
      gw[k] = a;
      if (ja.g_weights) ja.g_weights[(size_t) n * ja.K + k] = a;
    }
  }
  // ---- the weighting's backward (sp_knn.hip::sp_weights_backward_kernel, same expressions); `ja.logits`: the W weighting --
  // the ja.weights do not depend on the distances, its dense logit gradient is skgs_lbs_weights_backward
  float gf[SP_MAXF];
#pragma unroll
  for (int c = 0; c < SP_MAXF; ++c) gf[c] = 0.f;
  if (!ja.logits) {
    float dd[SP_MAXK];
#pragma unroll
    for (int k = 0; k < SP_MAXK; ++k) dd[k] = k < ja.K ? ja.nn_dist[(size_t) n * ja.K + k] : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SP_MAXK; ++k)
      if (k < ja.K) dot += ww[k] * gw[k];
    float sum = 0.f;
    if (ja.radius_raw)
#pragma unroll
      for (int k = 0; k < SP_MAXK; ++k)
        if (k < ja.K) {
          const float r = expf(ja.radius_raw[jj[k]]);
          float e = expf(-dd[k] / (2.f * (r * r)));
          if (ja.kweight_raw) e = e * (1.0f / (1.0f + expf(-ja.kweight_raw[jj[k]])));
          sum += e + 1e-7f;
        }
    float fc[SP_MAXF];
#pragma unroll
    for (int c = 0; c < SP_MAXF; ++c) fc[c] = (F > 0 && c < F) ? ja.feature[(size_t) n * F + c] : 0.f;
#pragma unroll
    for (int k = 0; k < SP_MAXK; ++k)
      if (k < ja.K) {
        const int j = jj[k];
        float g_d, rterm = 0.f, kterm = 0.f;
        if (ja.radius_raw) {
          const float r   = expf(ja.radius_raw[j]);
          const float e   = expf(-dd[k] / (2.f * (r * r)));
          const float sk  = ja.kweight_raw ? 1.0f / (1.0f + expf(-ja.kweight_raw[j])) : 1.f;
          const float g_u = (gw[k] - dot) / sum;
          const float g_e = g_u * sk;
          g_d   = g_e * e * (-1.f / (2.f * (r * r)));
          rterm = g_e * e * (dd[k] / (r * r * r));
          if (ja.kweight_raw) kterm = g_u * e;
        } else {
          g_d = -(ww[k] * (gw[k] - dot)) / ja.temperature;
        }
#pragma unroll
        for (int c = 0; c < SP_MAXF; ++c)
          if (F > 0 && c < F) gf[c] += g_d * 2.f * (fc[c] - ja.sp_feature[(size_t) j * F + c]);
        float* vp = ja.V + ((size_t) n * ja.K + k) * 4;
        *reinterpret_cast<float4*>(vp) = make_float4(g_d, rterm, kterm, 0.f);
      }
  }
  if (ja.g_feature)
#pragma unroll
    for (int c = 0; c < SP_MAXF; ++c)
      if (F > 0 && c < F) ja.g_feature[(size_t) n * F + c] = gf[c];
}

}  // namespace
}  // namespace skgs
