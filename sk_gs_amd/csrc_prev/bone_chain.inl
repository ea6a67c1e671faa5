// bone_chain.inl -- the kinematic chain (joint rotations -> bone transforms) and its backward as device functions run by
// ONE workgroup on a caller-provided LDS scratch: the bodies of bone_chain.hip's two kernels, also called from the deform
// network's launches (mlp_fused.hip: the chain follows the heads in workgroup 0 of the forward launch; its backward runs
// in every workgroup's prologue of the backward launch).  Math and references: bone_chain.hip.
#pragma once
#include "skgs_common.h"

namespace skgs {
namespace chain {

struct Q4 {
  float x, y, z, w;
};
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ Q4 qmul(const Q4& a, const Q4& b) {  // Hamilton product, xyzw
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
      a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ Q4 qconj(const Q4& a) { return {-a.x, -a.y, -a.z, a.w}; }
__device__ __forceinline__ Q4 qnormalize(const Q4& a) {
  const float n = fmaxf(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w), 1e-12f);
  return {a.x / n, a.y / n, a.z / n, a.w / n};
}
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// p + w*uv + v x uv, uv = 2 v x p  (lie.h:59-64)
__device__ __forceinline__ V3 qrot(const Q4& q, const V3& p) {
  const V3 v  = {q.x, q.y, q.z};
  V3 uv       = cross(v, p);
  uv          = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  const V3 c  = cross(v, uv);
  return {p.x + q.w * uv.x + c.x, p.y + q.w * uv.y + c.y, p.z + q.w * uv.z + c.z};
}
// gradient of (R(q) p) . g  w.r.t. q (as a polynomial in q) and w.r.t. p (= R(q)^T g)
__device__ __forceinline__ Q4 qrot_grad_q(const Q4& q, const V3& p, const V3& g) {
  const V3 v   = {q.x, q.y, q.z};
  const V3 pxg = cross(p, g), vxp = cross(v, p);
  const float vdp = dot(v, p), gdv = dot(g, v), gdp = dot(g, p);
  return {2.f * q.w * pxg.x + 2.f * (vdp * g.x + gdv * p.x - 2.f * gdp * v.x),
      2.f * q.w * pxg.y + 2.f * (vdp * g.y + gdv * p.y - 2.f * gdp * v.y),
      2.f * q.w * pxg.z + 2.f * (vdp * g.z + gdv * p.z - 2.f * gdp * v.z), 2.f * dot(g, vxp)};
}
__device__ __forceinline__ V3 qrot_T(const Q4& q, const V3& g) { return qrot(qconj(q), g); }

constexpr int LOC_F = 12;  // staged per-bone local data: q[4] (unit), tL[3], j[3], |raw|, pad

struct ChainArgs {
  int M, root, num_levels;
  const int32_t* parents;
  const int32_t* level_nodes;
  const int32_t* level_start;
  const float* sk_r_raw;     // [M,4] (may be an LDS address)
  const float* joints;       // [M,3]
  const float* global_T;     // [7] (or a [frames,7] table with frame_index), may be NULL
  const int32_t* frame_index;
  float* bone_T;             // forward: [M,7] written (NULL: not written)
  float* chain_A;            // forward: [M,7] written if not NULL; backward: read
  const float* g_bone_T;     // backward: [M,7]
  float* g_sk_r_raw;         // backward: [M,4] written if not NULL (global)
  float* g_joints;           // backward: [M,3] written if not NULL
  float* g_global_T;         // backward: [7] (row frame_index of a table) written if not NULL
  float* global_T_row;       // NULL, or 7 floats the caller owns: the forward leaves the frame's row of global_T there and the backward
                             // of the same frame reads it from there -- no load that waits for frame_index[0] first
};

// Every global load of a chain pass, issued at once into registers (bone i = threadIdx.x): the bodies below then run on
// LDS and registers only.  Without it a pass is a sequence of dependent round trips (frame index -> global transform,
// skeleton staging, chain transforms, incoming gradient): 11 us of the fused backward launch's prologue, 4 us at the end of
// the fused forward.  valid only if the skeleton fits one bone per thread; otherwise the bodies load for themselves.
struct Prefetch {
  bool valid, has_raw;
  float raw[4], j[3], a[7], g[7];
  float gT;  // word threadIdx.x (< 7) of the frame's global transform
  int par, ln, ls;
};
__device__ __forceinline__ Prefetch prefetch(const ChainArgs& c, bool want_raw, bool backward) {
  Prefetch p;
  p.valid   = c.M <= (int) blockDim.x && c.num_levels + 1 <= (int) blockDim.x;
  p.has_raw = want_raw;
  if (!p.valid) return p;
  const int i = threadIdx.x;
  const float* gTp = c.global_T;
  if (backward && gTp && c.global_T_row) {
    gTp = c.global_T_row;  // (the forward's copy of the row: one round trip instead of index -> row)
  } else if (gTp && c.frame_index) {
    gTp += 7 * (size_t) c.frame_index[0];
  }
  p.gT = (gTp && i < 7) ? gTp[i] : 0.f;
  p.ls = i <= c.num_levels ? c.level_start[i] : 0;
  p.par = p.ln = 0;
  if (i < c.M) {
    if (want_raw) p.raw[0] = c.sk_r_raw[4 * i], p.raw[1] = c.sk_r_raw[4 * i + 1], p.raw[2] = c.sk_r_raw[4 * i + 2], p.raw[3] = c.sk_r_raw[4 * i + 3];
    p.j[0] = c.joints[3 * i], p.j[1] = c.joints[3 * i + 1], p.j[2] = c.joints[3 * i + 2];
    p.par = c.parents[i], p.ln = c.level_nodes[i];
    if (backward) {
#pragma unroll
      for (int e = 0; e < 7; ++e) p.a[e] = c.chain_A[7 * i + e], p.g[e] = c.g_bone_T[7 * i + e];
    }
  }
  return p;
}

// Per-bone local transform L_i = (j + R(q)(-j), q) and the topology, staged in LDS by all threads at once: the
// level loops then touch LDS only.  (Reading parents / level_nodes / raw / joints from global memory inside the loop
// made every tree level a chain of three dependent global loads: 13 us for a 20-bone skeleton.)
struct Staged {
  const float* loc;  // [M][LOC_F]
  const int* par;    // [M]
  const int* ln;     // [M]
  const int* ls;     // [num_levels + 1]
};
template <bool PRE>
__device__ __forceinline__ Staged stage_skeleton(float* s_base, const ChainArgs& c, const Prefetch& pf) {
#pragma clang fp contract(off)
  const int M = c.M, nt = blockDim.x;
  constexpr bool pre = PRE;
  float* loc = s_base;
  int* par   = reinterpret_cast<int*>(s_base + (size_t) M * LOC_F);
  int* ln    = par + M;
  int* ls    = ln + M;
  for (int i = threadIdx.x; i < M; i += nt) {  // (pre: one pass, i = threadIdx.x)
    const bool pr = pre && pf.has_raw;
    const float raw[4] = {pr ? pf.raw[0] : c.sk_r_raw[4 * i], pr ? pf.raw[1] : c.sk_r_raw[4 * i + 1],
        pr ? pf.raw[2] : c.sk_r_raw[4 * i + 2], (pr ? pf.raw[3] : c.sk_r_raw[4 * i + 3]) + 1.0f};
    const float nraw   = fmaxf(sqrtf(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2] + raw[3] * raw[3]), 1e-12f);
    const Q4 q  = {raw[0] / nraw, raw[1] / nraw, raw[2] / nraw, raw[3] / nraw};
    V3 j;
    if constexpr (pre) j = V3{pf.j[0], pf.j[1], pf.j[2]};
    else j = V3{c.joints[3 * i], c.joints[3 * i + 1], c.joints[3 * i + 2]};
    const V3 rj = qrot(q, {-j.x, -j.y, -j.z});
    float* o    = loc + (size_t) i * LOC_F;
    o[0] = q.x, o[1] = q.y, o[2] = q.z, o[3] = q.w;
    o[4] = j.x + rj.x, o[5] = j.y + rj.y, o[6] = j.z + rj.z;
    o[7] = j.x, o[8] = j.y, o[9] = j.z, o[10] = nraw, o[11] = 0.f;
    if constexpr (pre) par[i] = pf.par, ln[i] = pf.ln;
    else par[i] = c.parents[i], ln[i] = c.level_nodes[i];
  }
  for (int l = threadIdx.x; l <= c.num_levels; l += nt) {
    if constexpr (pre) ls[l] = pf.ls;
    else ls[l] = c.level_start[l];
  }
  return {loc, par, ln, ls};
}
// the frame's global transform (7 floats) into LDS: from the prefetch, or loaded here
template <bool PRE>
__device__ __forceinline__ void stage_global_T(float* s_gT, const ChainArgs& c, const Prefetch& pf) {
  if (threadIdx.x < 7) {
    float v = 0.f;
    if constexpr (PRE) {
      v = pf.gT;
    } else if (c.global_T) {
      const float* gTp = c.global_T;
      if (c.frame_index) gTp += 7 * (size_t) c.frame_index[0];  // row of a [frames, 7] table, chosen on the device
      v = gTp[threadIdx.x];
    }
    s_gT[threadIdx.x] = v;
  }
}
__host__ __device__ inline size_t staged_floats(int M, int num_levels) { return (size_t) M * LOC_F + 2 * (size_t) M + num_levels + 2; }
__host__ __device__ inline size_t forward_scratch_floats(int M, int num_levels) { return 7 * (size_t) M + 8 + staged_floats(M, num_levels); }
__host__ __device__ inline size_t backward_scratch_floats(int M, int num_levels) { return 14 * (size_t) M + 16 + staged_floats(M, num_levels); }

// ---- forward: s_mem = [M][7] chain transforms A | global transform [8] | staged skeleton.  Every thread of the workgroup
// must call it.  pf: the loads issued earlier (NULL: loaded here).
template <bool PRE>
__device__ __forceinline__ void forward_body_t(float* s_mem, const ChainArgs& c, const Prefetch& pf) {
#pragma clang fp contract(off)
  const int M = c.M, nt = blockDim.x, tid = threadIdx.x;
  float* s_A = s_mem;
  float* s_gT = s_mem + 7 * (size_t) M;
  const float* global_T = c.global_T ? s_gT : nullptr;
  stage_global_T<PRE>(s_gT, c, pf);
  if (c.global_T && c.global_T_row && tid < 7) c.global_T_row[tid] = s_gT[tid];  // (this thread staged that word)
  const Staged sk = stage_skeleton<PRE>(s_gT + 8, c, pf);
  if (tid == 0) {
    float* a = s_A + 7 * c.root;
    a[0] = a[1] = a[2] = a[3] = a[4] = a[5] = 0.f, a[6] = 1.f;
  }
  __syncthreads();
  for (int lv = 1; lv < c.num_levels; ++lv) {
    for (int k = sk.ls[lv] + tid; k < sk.ls[lv + 1]; k += nt) {
      const int i = sk.ln[k], p = sk.par[i];
      const float* lo = sk.loc + (size_t) i * LOC_F;
      const Q4 q  = {lo[0], lo[1], lo[2], lo[3]};
      const V3 tL = {lo[4], lo[5], lo[6]};
      const float* ap = s_A + 7 * p;
      const Q4 qp = {ap[3], ap[4], ap[5], ap[6]};
      const V3 rt = qrot(qp, tL);
      const Q4 qa = qnormalize(qmul(qp, q));
      float* a    = s_A + 7 * i;
      a[0] = ap[0] + rt.x, a[1] = ap[1] + rt.y, a[2] = ap[2] + rt.z;
      a[3] = qa.x, a[4] = qa.y, a[5] = qa.z, a[6] = qa.w;
    }
    __syncthreads();
  }
  Q4 qg = {0.f, 0.f, 0.f, 1.f};
  V3 tg = {0.f, 0.f, 0.f};
  if (global_T) {
    qg = qnormalize({global_T[3], global_T[4], global_T[5], global_T[6]});
    tg = {global_T[0], global_T[1], global_T[2]};
  }
  for (int i = tid; i < M; i += nt) {
    const float* a = s_A + 7 * i;
    const Q4 qa    = {a[3], a[4], a[5], a[6]};
    float* o       = c.bone_T + 7 * i;
    if (global_T) {
      const V3 rt = qrot(qg, {a[0], a[1], a[2]});
      const Q4 qt = qnormalize(qmul(qg, qa));
      o[0] = tg.x + rt.x, o[1] = tg.y + rt.y, o[2] = tg.z + rt.z, o[3] = qt.x, o[4] = qt.y, o[5] = qt.z, o[6] = qt.w;
    } else {
#pragma unroll
      for (int e = 0; e < 7; ++e) o[e] = a[e];
    }
    if (c.chain_A) {
#pragma unroll
      for (int e = 0; e < 7; ++e) c.chain_A[7 * i + e] = a[e];
    }
  }
}

// ---- backward, in two phases so that a caller can issue other loads in between:
//   backward_stage : every input into LDS (from the prefetch or loaded here) and the gradient of A_i from T_i = G o A_i
//   backward_levels: the tree walk.  Only the parent accumulation gA_p += f(gA_i) is a dependent chain (deepest level
//                    first); the local gradients (joint rotation, joint position) of ALL bones are formed in one parallel
//                    pass once every gA is final -- inside the level loop they were two thirds of its instructions, on the
//                    critical path of the fused backward launch (11 us -> 6).
// s_mem = A [M][7] | gA [M][7] | gG [8] | global transform [8] | staged skeleton.  Every thread of the workgroup must call
// both.  s_g_raw: [M][4] destination of the raw rotations' gradient (LDS or global, never NULL); `write_global`: this
// workgroup also writes c.g_sk_r_raw / c.g_joints / c.g_global_T (the others only need s_g_raw).
template <bool PRE>
__device__ __forceinline__ void backward_stage_t(float* s_mem, const ChainArgs& ca, const Prefetch& pf) {
#pragma clang fp contract(off)
  const int M = ca.M, nt = blockDim.x;
  float* s_A  = s_mem;          // [M][7]
  float* s_gA = s_mem + 7 * M;  // [M][7]
  float* s_gG = s_mem + 14 * (size_t) M;  // [7] (+ pad)
  float* s_gT = s_gG + 8;                 // [7] (+ pad): the frame's global transform
  const int tid = threadIdx.x;
  constexpr bool pre = PRE;
  stage_global_T<PRE>(s_gT, ca, pf);
  stage_skeleton<PRE>(s_mem + 14 * (size_t) M + 16, ca, pf);
  if constexpr (pre) {
    if (tid < M) {
#pragma unroll
      for (int e = 0; e < 7; ++e) s_A[7 * tid + e] = pf.a[e];
    }
  } else {
    for (int i = tid; i < 7 * M; i += nt) s_A[i] = ca.chain_A[i];
  }
  if (tid < 7) s_gG[tid] = 0.f;
  __syncthreads();
  const float* global_T = ca.global_T ? s_gT : nullptr;
  Q4 qg = {0.f, 0.f, 0.f, 1.f};
  if (global_T) qg = qnormalize({global_T[3], global_T[4], global_T[5], global_T[6]});
  // T_i = G o A_i
  for (int i = tid; i < M; i += nt) {
    float g[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) {
      if constexpr (pre) g[e] = pf.g[e];
      else g[e] = ca.g_bone_T[7 * i + e];
    }
    const V3 gt    = {g[0], g[1], g[2]};
    const Q4 gq    = {g[3], g[4], g[5], g[6]};
    float* ga      = s_gA + 7 * i;
    if (global_T) {
      const float* a = s_A + 7 * i;
      const V3 tA    = {a[0], a[1], a[2]};
      const Q4 qA    = {a[3], a[4], a[5], a[6]};
      const V3 gtA   = qrot_T(qg, gt);
      const Q4 gqA   = qmul(qconj(qg), gq);          // d(qg * qA)/dqA ^T
      const Q4 gqG1  = qrot_grad_q(qg, tA, gt);      // through R(qg) tA
      const Q4 gqG2  = qmul(gq, qconj(qA));          // d(qg * qA)/dqg ^T
      ga[0] = gtA.x, ga[1] = gtA.y, ga[2] = gtA.z, ga[3] = gqA.x, ga[4] = gqA.y, ga[5] = gqA.z, ga[6] = gqA.w;
      atomicAdd(&s_gG[0], gt.x), atomicAdd(&s_gG[1], gt.y), atomicAdd(&s_gG[2], gt.z);
      atomicAdd(&s_gG[3], gqG1.x + gqG2.x), atomicAdd(&s_gG[4], gqG1.y + gqG2.y);
      atomicAdd(&s_gG[5], gqG1.z + gqG2.z), atomicAdd(&s_gG[6], gqG1.w + gqG2.w);
    } else {
#pragma unroll
      for (int c = 0; c < 7; ++c) ga[c] = g[c];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void backward_levels(float* s_mem, const ChainArgs& ca, float* s_g_raw, bool write_global) {
#pragma clang fp contract(off)
  const int M = ca.M, nt = blockDim.x, root = ca.root, num_levels = ca.num_levels, tid = threadIdx.x;
  float* g_global_T     = write_global ? ca.g_global_T : nullptr;
  float* g_joints       = write_global ? ca.g_joints : nullptr;
  float* g_sk_r_raw     = s_g_raw;
  float* g_raw_global   = (write_global && ca.g_sk_r_raw != s_g_raw) ? ca.g_sk_r_raw : nullptr;
  if (ca.frame_index && g_global_T) g_global_T += 7 * (size_t) ca.frame_index[0];
  float* s_A  = s_mem;
  float* s_gA = s_mem + 7 * M;
  float* s_gG = s_mem + 14 * (size_t) M;
  float* s_gT = s_gG + 8;
  const float* global_T = ca.global_T ? s_gT : nullptr;
  float* sb = s_mem + 14 * (size_t) M + 16;  // the staged skeleton (layout of stage_skeleton)
  const Staged sk = {sb, reinterpret_cast<int*>(sb + (size_t) M * LOC_F), reinterpret_cast<int*>(sb + (size_t) M * LOC_F) + M,
      reinterpret_cast<int*>(sb + (size_t) M * LOC_F) + 2 * M};
  // A_i = A_p o L_i, deepest level first: the parent's share of gA_i
  for (int lv = num_levels - 1; lv >= 1; --lv) {
    for (int k = sk.ls[lv] + tid; k < sk.ls[lv + 1]; k += nt) {
      const int i = sk.ln[k], p = sk.par[i];
      const float* lo = sk.loc + (size_t) i * LOC_F;
      const Q4 q      = {lo[0], lo[1], lo[2], lo[3]};
      const V3 tL     = {lo[4], lo[5], lo[6]};
      const float* ap = s_A + 7 * p;
      const Q4 qp = {ap[3], ap[4], ap[5], ap[6]};
      const float* ga = s_gA + 7 * i;
      const V3 gtA = {ga[0], ga[1], ga[2]};
      const Q4 gqA = {ga[3], ga[4], ga[5], ga[6]};
      const Q4 gqp1 = qrot_grad_q(qp, tL, gtA);
      const Q4 gqp2 = qmul(gqA, qconj(q));
      float* gp     = s_gA + 7 * p;
      atomicAdd(&gp[0], gtA.x), atomicAdd(&gp[1], gtA.y), atomicAdd(&gp[2], gtA.z);
      atomicAdd(&gp[3], gqp1.x + gqp2.x), atomicAdd(&gp[4], gqp1.y + gqp2.y);
      atomicAdd(&gp[5], gqp1.z + gqp2.z), atomicAdd(&gp[6], gqp1.w + gqp2.w);
    }
    __syncthreads();
  }
  // every gA is final: the local transform L_i = (j + R(q)(-j), q) of all bones at once
  for (int i = tid; i < M; i += nt) {
    if (i == root) {  // the root's own rotation is replaced by the identity: no gradient
      g_sk_r_raw[4 * i] = g_sk_r_raw[4 * i + 1] = g_sk_r_raw[4 * i + 2] = g_sk_r_raw[4 * i + 3] = 0.f;
      if (g_joints) g_joints[3 * i] = g_joints[3 * i + 1] = g_joints[3 * i + 2] = 0.f;
      continue;
    }
    const int p = sk.par[i];
    const float* lo  = sk.loc + (size_t) i * LOC_F;
    const Q4 q       = {lo[0], lo[1], lo[2], lo[3]};
    const V3 mj      = {-lo[7], -lo[8], -lo[9]};
    const float nraw = lo[10];
    const float* ap = s_A + 7 * p;
    const Q4 qp = {ap[3], ap[4], ap[5], ap[6]};
    const float* ga = s_gA + 7 * i;
    const V3 gtA = {ga[0], ga[1], ga[2]};
    const Q4 gqA = {ga[3], ga[4], ga[5], ga[6]};
    const V3 gtL = qrot_T(qp, gtA);
    Q4 gq        = qmul(qconj(qp), gqA);
    const Q4 gq2 = qrot_grad_q(q, mj, gtL);
    gq           = {gq.x + gq2.x, gq.y + gq2.y, gq.z + gq2.z, gq.w + gq2.w};
    if (g_joints) {
      const V3 rtg = qrot_T(q, gtL);  // d(R(q)(-j))/dj = -R(q)
      g_joints[3 * i] = gtL.x - rtg.x, g_joints[3 * i + 1] = gtL.y - rtg.y, g_joints[3 * i + 2] = gtL.z - rtg.z;
    }
    // raw -> unit quaternion
    const float d = q.x * gq.x + q.y * gq.y + q.z * gq.z + q.w * gq.w;
    g_sk_r_raw[4 * i]     = (gq.x - q.x * d) / nraw;
    g_sk_r_raw[4 * i + 1] = (gq.y - q.y * d) / nraw;
    g_sk_r_raw[4 * i + 2] = (gq.z - q.z * d) / nraw;
    g_sk_r_raw[4 * i + 3] = (gq.w - q.w * d) / nraw;
  }
  if (tid == 0 && g_global_T) {
    if (global_T) {
      const Q4 qg   = qnormalize({global_T[3], global_T[4], global_T[5], global_T[6]});
      const float n = fmaxf(sqrtf(global_T[3] * global_T[3] + global_T[4] * global_T[4] + global_T[5] * global_T[5] +
                                  global_T[6] * global_T[6]), 1e-12f);
      const float d = qg.x * s_gG[3] + qg.y * s_gG[4] + qg.z * s_gG[5] + qg.w * s_gG[6];
      g_global_T[0] = s_gG[0], g_global_T[1] = s_gG[1], g_global_T[2] = s_gG[2];
      g_global_T[3] = (s_gG[3] - qg.x * d) / n, g_global_T[4] = (s_gG[4] - qg.y * d) / n;
      g_global_T[5] = (s_gG[5] - qg.z * d) / n, g_global_T[6] = (s_gG[6] - qg.w * d) / n;
    } else {
#pragma unroll
      for (int c = 0; c < 7; ++c) g_global_T[c] = 0.f;
    }
  }
  if (g_raw_global) {
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * M; i += nt) g_raw_global[i] = s_g_raw[i];
  }
}
// pf by value / reference into force-inlined templates: behind a pointer the struct lived in scratch memory
__device__ __forceinline__ void backward_stage(float* s_mem, const ChainArgs& ca, const Prefetch& pf) {
  if (pf.valid) backward_stage_t<true>(s_mem, ca, pf);
  else backward_stage_t<false>(s_mem, ca, pf);
}
__device__ __forceinline__ void backward_body(float* s_mem, const ChainArgs& ca, float* s_g_raw, bool write_global,
    const Prefetch& pf) {
  backward_stage(s_mem, ca, pf);
  backward_levels(s_mem, ca, s_g_raw, write_global);
}
__device__ __forceinline__ void forward_body(float* s_mem, const ChainArgs& c, const Prefetch& pf) {
  if (pf.valid) forward_body_t<true>(s_mem, c, pf);
  else forward_body_t<false>(s_mem, c, pf);
}

}  // namespace chain
}  // namespace skgs
