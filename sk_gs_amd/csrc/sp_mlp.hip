// sp_mlp.hip -- the deform network of the SUPERPOINT stage on MFMA row blocks.
//
// Stage `sp` (networks/sk_gs.py:830-856; 30 k of the reference's 80 k default steps, exps/default.yaml:12-19) evaluates
// `sp_deform_net` = DeformNetwork (sk_gs.py:209-315, is_blender=True: exps/default.yaml:31) on the M = 512 superpoints
// (exps/default.yaml:25) every step:
//     t_emb = timenet(freq(t, 6))            Linear(13,256) ReLU Linear(256,30)                       (:250-253,297-299)
//     h     = [freq(x, 10) | t_emb]          63 + 30 = 93 columns                                     (:300-301)
//     8 x   h = relu(linear[i](h)); after layer 4: h = [freq(x) | t_emb | h]  (the input goes IN FRONT) (:302-306)
//     d_xyz = gaussian_warp(h), scaling = gaussian_scaling(h), rotation = gaussian_rotation(h)         (:308-310)
// and sp_stage normalises `rotation + [0,0,0,1]` (:847).  In torch that is ~40 launches forward and ~90 backward.
//
// 512 rows x 256 x 256 per layer is where the matrix cores pay: 0.5 GFLOP forward, 1 GFLOP backward, all
// fp32 (v_mfma_f32_16x16x4_f32: exact f32, one rounding per product -- MI355X_MICROARCH "FP32-input MFMA").
//
//   forward    ONE launch.  The ROWS are independent, so a workgroup takes 16 superpoints through the WHOLE network: no
//              exchange between workgroups, no barrier across the grid.  4 waves; wave w owns output features [64w, 64w+64) of
//              every layer as four 16x16 tiles.  The 16 x 256 activation block lives in LDS (two buffers), the weights
//              stream from L2 (2.3 MB for all layers: every XCD's 4 MB L2 holds them after the first touch).
//              K-permutation: lane (j, q) loads ONE float4 of weight row j -- columns 16s + 4q .. +3 -- per tile and step and
//              feeds component t to MFMA t, so MFMA t contracts k in {16s + 4q + t}; the activation operand is the matching
//              float4 from LDS.  1 ds_read_b128 + 4 global_load_dwordx4 per 16 MFMAs.
//   backward A the same row blocks walk back: gZ = gY * (Y > 0), gX = gZ W.  Column interleave: lane (j, q) loads one float4 of
//              weight row o = 16s + 4q + t at input columns n0 + 4j .. +3 and component c goes to tile c, so tile c holds
//              input features {n0 + 4j + c}: again 4 x dwordx4 per 16 MFMAs, and every lane ends with float4s of consecutive
//              features (16-byte stores).  Every layer's gZ is saved for launch B.  The launch occupies M / 16 = 32 CUs; its
//              other workgroups run an optimizer piece (skgs_adam_range, as skgs_skeleton_backward does).
//   backward B all weight gradients gW_l = gZ_l^T X_l (K = the 512 rows) as 64 x 64 output tiles on the whole chip: 132
//              workgroups, the rows split over the 4 waves and summed through LDS; bias gradients = column sums of gZ_l; the
//              time network's backward in the workgroup that finishes last (d loss / d t_emb = gb_0 W_0[:, 63:93] + gb_5
//              W_5[:, 63:93]: t_emb is the same for every row, so its gradient needs only the bias gradients).
// No gradient w.r.t. the superpoint positions: every caller detaches them (sk_gs.py:746-748,845).
#include <algorithm>
#include <cstdint>

#include "adam_update.h"
#include "skgs_common.h"

namespace skgs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SPW    = 256;  // layer width
constexpr int SPD    = 8;    // hidden layers
constexpr int SKIP   = 4;    // after this layer the encoded input is concatenated in front
constexpr int PDEG   = 10, TDEG = 6;
constexpr int PDIM   = 3 * (1 + 2 * PDEG);   // 63
constexpr int TDIM   = 1 + 2 * TDEG;         // 13
constexpr int THID   = 256, TOUT = 30;
constexpr int IN0    = PDIM + TOUT;          // 93
constexpr int IN0P   = 96;                   // padded row of the saved encoded input
constexpr int ROWS   = 16;                   // superpoints per workgroup = the MFMA tile's rows
constexpr int NT     = 256;                  // 4 waves
constexpr int PITCH  = SPW + 4;              // LDS row pitch of an activation block (floats)
constexpr int XPITCH = IN0P + 4;
constexpr int NOUT   = 10;                   // d_xyz 3 | d_rotation 4 | d_scaling 3

struct __attribute__((packed, aligned(4))) f4u {  // a float4 at 4-byte alignment (weight rows of 93 / 349 floats)
  float x, y, z, w;
};
__device__ __forceinline__ float4 ldg4(const float* p) {
  const f4u v = *reinterpret_cast<const f4u*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ldg4_guard(const float* p, int valid) {  // elements [0, valid) exist
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid >= 4) return ldg4(p);
  if (valid > 0) r.x = p[0];
  if (valid > 1) r.y = p[1];
  if (valid > 2) r.z = p[2];
  return r;
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// saved by the forward for the backward (floats), Mp = rows rounded up to 16
struct SavedView {
  float* x0;    // [Mp][IN0P]   encoded input (columns 93..95 zero)
  float* Y;     // [SPD][Mp][SPW] post-ReLU activations
  float* rawq;  // [Mp][4]      raw rotation head (before + [0,0,0,1] and the normalisation)
  float* temb;  // [16]         freq(t) (13 used)
  float* thid;  // [THID]       hidden layer of the time network (post-ReLU)
  float* tout;  // [32]         its output (30 used)
};
__host__ __device__ inline int pad_rows(int M) { return (M + ROWS - 1) / ROWS * ROWS; }
__host__ __device__ inline size_t saved_floats(int M) {
  const size_t Mp = pad_rows(M);
  return Mp * IN0P + (size_t) SPD * Mp * SPW + Mp * 4 + 16 + THID + 32;
}
__host__ __device__ inline SavedView saved_view(void* base, int M) {
  const size_t Mp = pad_rows(M);
  SavedView v;
  float* p = reinterpret_cast<float*>(base);
  v.x0 = p, p += Mp * IN0P;
  v.Y = p, p += (size_t) SPD * Mp * SPW;
  v.rawq = p, p += Mp * 4;
  v.temb = p, p += 16;
  v.thid = p, p += THID;
  v.tout = p;
  return v;
}
// backward workspace: ticket (256 B) | GH [Mp][16] head cotangents (10 used) | GZ [SPD][Mp][SPW]
struct WorkView {
  unsigned* ticket;
  float* GH;
  float* GZ;
};
__host__ __device__ inline size_t work_bytes(int M) {
  const size_t Mp = pad_rows(M);
  return 256 + (Mp * 16 + (size_t) SPD * Mp * SPW) * 4;
}
__host__ __device__ inline WorkView work_view(void* base, int M) {
  const size_t Mp = pad_rows(M);
  WorkView v;
  v.ticket = reinterpret_cast<unsigned*>(base);
  v.GH     = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + 256);
  v.GZ     = v.GH + Mp * 16;
  return v;
}

struct SideAdam {
  const AdamTensor* tensors;
  int n;
  int64_t c0, c1;
  double beta1, beta2;
  float eps;
  const AdamState* state;
  int after_advance;
};

// input width / weight row stride / column offset of the hidden part of layer l
__host__ __device__ inline int layer_ld(int l) { return l == 0 ? IN0 : (l == SKIP + 1 ? IN0 + SPW : SPW); }
__host__ __device__ inline int layer_hofs(int l) { return l == SKIP + 1 ? IN0 : 0; }

// acc[c] (tile c = output features 64 wave + 16 c + j) += A[16 x 16 nsteps] W^T, A from LDS (row pitch pa), W rows at stride ldw
// starting at column kofs; `kvalid` = number of valid weight columns from kofs (the tail of the 93-wide input is guarded).
template <bool GUARD>
__device__ __forceinline__ void gemm_fwd(f32x4 (&acc)[4], const float* sA, int pa, const float* __restrict__ W, int ldw, int kofs,
    int nsteps, int kvalid, int wave, int lane) {
  const int j = lane & 15, q = lane >> 4;
  const float* arow = sA + j * pa + 4 * q;
  const float* wrow[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) wrow[c] = W + (size_t) (64 * wave + 16 * c + j) * ldw + kofs + 4 * q;
  float4 b[2][4];
  auto fetch = [&](int s, float4 (&dst)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = GUARD ? ldg4_guard(wrow[c] + 16 * s, kvalid - (16 * s + 4 * q)) : ldg4(wrow[c] + 16 * s);
  };
  fetch(0, b[0]);
  for (int s = 0; s < nsteps; s += 2) {
    if (s + 1 < nsteps) fetch(s + 1, b[1]);
    {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * s);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = mfma4(a.x, b[0][c].x, acc[c]);
        acc[c] = mfma4(a.y, b[0][c].y, acc[c]);
        acc[c] = mfma4(a.z, b[0][c].z, acc[c]);
        acc[c] = mfma4(a.w, b[0][c].w, acc[c]);
      }
    }
    if (s + 2 < nsteps) fetch(s + 2, b[0]);
    if (s + 1 < nsteps) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * (s + 1));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = mfma4(a.x, b[1][c].x, acc[c]);
        acc[c] = mfma4(a.y, b[1][c].y, acc[c]);
        acc[c] = mfma4(a.z, b[1][c].z, acc[c]);
        acc[c] = mfma4(a.w, b[1][c].w, acc[c]);
      }
    }
  }
}

// acc[c] (tile c = INPUT features n0 + 4 j + c of a wave's 64: n0 = 64 wave) += gZ[16 x 256] W[:, kofs + n0 ...]: the contraction
// runs over the layer's 256 output features o = 16 s + 4 q + t, one float4 of weight row o per (s, t)
__device__ __forceinline__ void gemm_bwd(f32x4 (&acc)[4], const float* sA, int pa, const float* __restrict__ W, int ldw, int kofs,
    int wave, int lane) {
  const int j = lane & 15, q = lane >> 4;
  const float* arow = sA + j * pa + 4 * q;
  const float* wcol = W + (size_t) (4 * q) * ldw + kofs + 64 * wave + 4 * j;
  float4 b[2][4];
  auto fetch = [&](int s, float4 (&dst)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[t] = ldg4(wcol + (size_t) (16 * s + t) * ldw);
  };
  fetch(0, b[0]);
  constexpr int nsteps = SPW / 16;
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 2) {
    fetch(s + 1, b[1]);
    {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * s);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0] = mfma4(av[t], b[0][t].x, acc[0]);
        acc[1] = mfma4(av[t], b[0][t].y, acc[1]);
        acc[2] = mfma4(av[t], b[0][t].z, acc[2]);
        acc[3] = mfma4(av[t], b[0][t].w, acc[3]);
      }
    }
    if (s + 2 < nsteps) fetch(s + 2, b[0]);
    {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * (s + 1));
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0] = mfma4(av[t], b[1][t].x, acc[0]);
        acc[1] = mfma4(av[t], b[1][t].y, acc[1]);
        acc[2] = mfma4(av[t], b[1][t].z, acc[2]);
        acc[3] = mfma4(av[t], b[1][t].w, acc[3]);
      }
    }
  }
}

struct NetPtrs {  // device copy of skgs_sp_net's pointers
  const float* points;
  const float* time;
  const float *tw1, *tb1, *tw2, *tb2;
  const float* W[SPD];
  const float* b[SPD];
  const float* head_w[3];  // warp (3), rotation (4), scaling (3): the order of the raw output row
  const float* head_b[3];
};
struct GradPtrs {
  float *tw1, *tb1, *tw2, *tb2;
  float* W[SPD];
  float* b[SPD];
  float* head_w[3];
  float* head_b[3];
};
__device__ __forceinline__ int head_of(int o, int& row) {  // raw output column -> (head, row of that head's matrix)
  if (o < 3) return row = o, 0;
  if (o < 7) return row = o - 3, 1;
  return row = o - 7, 2;
}

// =================================================================================================== forward
__global__ void __launch_bounds__(NT) sp_net_forward_kernel(int M, NetPtrs n, float* __restrict__ raw, float* __restrict__ bone_T,
    float* __restrict__ d_rot, float* __restrict__ d_scale, SavedView sv) {
  __shared__ __attribute__((aligned(16))) float s_x0[ROWS * XPITCH];
  __shared__ __attribute__((aligned(16))) float s_act[2][ROWS * PITCH];
  __shared__ float s_temb[16], s_thid[THID], s_tout[32];
  __shared__ __attribute__((aligned(16))) float s_head[4][ROWS][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * ROWS, Mp = pad_rows(M);
  // ---- time network (every workgroup: 13 -> 256 -> 30 is ~11 k multiply-adds)
  if (tid < TDIM) {
    const float t = n.time[0];
    float v = t;
    if (tid >= 1) {
      const int col = tid - 1;
      v = sinf(scalbnf(t, col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
    }
    s_temb[tid] = v;
  }
  __syncthreads();
  {
    float h = n.tb1[tid];
#pragma unroll
    for (int k = 0; k < TDIM; ++k) h += n.tw1[tid * TDIM + k] * s_temb[k];
    s_thid[tid] = fmaxf(h, 0.f);
  }
  __syncthreads();
  if (tid < TOUT * 8) {  // 8 lanes per output
    const int o = tid >> 3, part = tid & 7;
    float v = 0.f;
    for (int k = part; k < THID; k += 8) v += n.tw2[o * THID + k] * s_thid[k];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (part == 0) s_tout[o] = v + n.tb2[o];
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < 16) sv.temb[tid] = tid < TDIM ? s_temb[tid] : 0.f;
    sv.thid[tid] = s_thid[tid];
    if (tid < 32) sv.tout[tid] = tid < TOUT ? s_tout[tid] : 0.f;
  }
  // ---- encoded input of the 16 rows: [x | sin / cos(2^f x) ...] (freqencoder.cu:7-31) | t_emb | 0 0 0
  for (int e = tid; e < ROWS * IN0P; e += NT) {
    const int row = e / IN0P, c = e - row * IN0P;
    const int gr = min(r0 + row, M - 1);  // (rows beyond M repeat the last one: computed, never stored outside `saved`)
    float v = 0.f;
    if (c < 3) {
      v = n.points[3 * gr + c];
    } else if (c < PDIM) {
      const int col = c / 3 - 1, d = c % 3;
      v = sinf(scalbnf(n.points[3 * gr + d], col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
    } else if (c < IN0) {
      v = s_tout[c - PDIM];
    }
    s_x0[row * XPITCH + c]                 = v;
    sv.x0[(size_t) (r0 + row) * IN0P + c] = v;
  }
  __syncthreads();
  // ---- the eight layers
  const int j = lane & 15, q = lane >> 4;
  int cur = 0;
  for (int l = 0; l < SPD; ++l) {
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* W = n.W[l];
    const int ldw  = layer_ld(l);
    if (l == 0 || l == SKIP + 1) gemm_fwd<true>(acc, s_x0, XPITCH, W, ldw, 0, IN0P / 16, IN0, wave, lane);
    if (l > 0) gemm_fwd<false>(acc, s_act[cur], PITCH, W, ldw, layer_hofs(l), SPW / 16, SPW, wave, lane);
    float* out  = s_act[cur ^ 1];
    float* Yl   = sv.Y + ((size_t) l * Mp + r0) * SPW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int f    = 64 * wave + 16 * c + j;
      const float bb = n.b[l][f];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        const float v = fmaxf(acc[c][r] + bb, 0.f);
        out[row * PITCH + f]       = v;
        Yl[(size_t) row * SPW + f] = v;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  // ---- heads: raw[16 x 10] = h W_heads^T + b; the contraction split over the 4 waves (64 k each), summed through LDS
  {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int hrow;
    const int hd        = head_of(min(j, NOUT - 1), hrow);
    const float* wr     = n.head_w[hd] + (size_t) hrow * SPW + 64 * wave + 4 * q;
    const float* arow   = s_act[cur] + j * PITCH + 64 * wave + 4 * q;
    const bool real_col = j < NOUT;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 16 * s);
      float4 b       = *reinterpret_cast<const float4*>(wr + 16 * s);
      if (!real_col) b = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = mfma4(a.x, b.x, acc);
      acc = mfma4(a.y, b.y, acc);
      acc = mfma4(a.z, b.z, acc);
      acc = mfma4(a.w, b.w, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_head[wave][4 * q + r][j] = acc[r];
  }
  __syncthreads();
  if (tid < ROWS) {  // one thread per superpoint: the three raw outputs and the stage's epilogue (sk_gs.py:847)
    const int gr = r0 + tid;
    float o[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; ++c) {
      int hrow;
      const int hd = head_of(c, hrow);
      o[c] = ((s_head[0][tid][c] + s_head[1][tid][c]) + (s_head[2][tid][c] + s_head[3][tid][c])) + n.head_b[hd][hrow];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sv.rawq[(size_t) gr * 4 + c] = o[3 + c];
    if (gr < M) {
      if (raw)
#pragma unroll
        for (int c = 0; c < NOUT; ++c) raw[(size_t) gr * NOUT + c] = o[c];
      const float v[4] = {o[3], o[4], o[5], o[6] + 1.0f};
      const float nv   = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]), 1e-12f);  // F.normalize eps
      if (bone_T) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bone_T[(size_t) gr * 7 + c] = o[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) bone_T[(size_t) gr * 7 + 3 + c] = v[c] / nv;
      }
      if (d_rot)
#pragma unroll
        for (int c = 0; c < 4; ++c) d_rot[(size_t) gr * 4 + c] = v[c] / nv;
      if (d_scale)
#pragma unroll
        for (int c = 0; c < 3; ++c) d_scale[(size_t) gr * 3 + c] = o[7 + c];
    }
  }
}

// ================================================================================================ backward, launch A
__device__ __forceinline__ void side_adam_walk(const SideAdam& a, int wg, int n_side) {
  const int t256 = threadIdx.x, lane = threadIdx.x & 63;
  const AdamCoef k            = adam_coefficients(a.beta1, a.beta2, a.eps, a.state, a.after_advance != 0);
  const AdamTensorLanes desc  = adam_load_descriptors(a.tensors, a.n, lane);
  const int64_t first0        = lane < a.n ? a.tensors[lane].chunk0 : INT64_MAX;
  const int64_t n_chunks      = a.c1 - a.c0;
  const int64_t begin = a.c0 + n_chunks * wg / n_side, end = a.c0 + n_chunks * (wg + 1) / n_side;
  for (int64_t chunk = begin; chunk < end; chunk += 2) {
    const int ti0      = adam_owner(a.tensors, a.n, first0, lane, chunk);
    const AdamTensor T0 = ti0 < 64 ? adam_descriptor_of(desc, ti0) : a.tensors[ti0];
    if (chunk + 1 < end) {
      const int ti1      = adam_owner(a.tensors, a.n, first0, lane, chunk + 1);
      const AdamTensor T1 = ti1 < 64 ? adam_descriptor_of(desc, ti1) : a.tensors[ti1];
      adam_update_chunk2(T0, (chunk - T0.chunk0) * ADAM_CHUNK, T1, (chunk + 1 - T1.chunk0) * ADAM_CHUNK, t256, k);
    } else {
      adam_update_chunk(T0, (chunk - T0.chunk0) * ADAM_CHUNK, t256, k);
    }
  }
}

__global__ void __launch_bounds__(NT) sp_net_backward_rows_kernel(int M, int nblk, NetPtrs n, const float* __restrict__ g_bone_T,
    const float* __restrict__ g_d_rot, const float* __restrict__ g_d_scale, const float* __restrict__ g_raw, SavedView sv,
    WorkView wk, SideAdam side) {
  if ((int) blockIdx.x >= nblk) {  // the CUs the 32 row blocks leave idle: an optimizer piece
    side_adam_walk(side, (int) blockIdx.x - nblk, (int) gridDim.x - nblk);
    return;
  }
  __shared__ __attribute__((aligned(16))) float s_gz[2][ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float s_gh[ROWS][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * ROWS, Mp = pad_rows(M);
  // ---- cotangent of the raw output row [d_xyz 3 | rotation 4 | scaling 3]
  if (tid < ROWS) {
    const int gr = r0 + tid;
    float g[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) g[c] = 0.f;
    if (gr < M) {
      if (g_raw) {
#pragma unroll
        for (int c = 0; c < NOUT; ++c) g[c] = g_raw[(size_t) gr * NOUT + c];
      } else {
        // bone_T = [d_xyz | u], d_rot = u, u = v / |v|, v = rotation + [0,0,0,1]:  g_v = (g_u - u (u . g_u)) / |v|
        float gu[4] = {0.f, 0.f, 0.f, 0.f};
        if (g_bone_T) {
#pragma unroll
          for (int c = 0; c < 3; ++c) g[c] = g_bone_T[(size_t) gr * 7 + c];
#pragma unroll
          for (int c = 0; c < 4; ++c) gu[c] = g_bone_T[(size_t) gr * 7 + 3 + c];
        }
        if (g_d_rot)
#pragma unroll
          for (int c = 0; c < 4; ++c) gu[c] += g_d_rot[(size_t) gr * 4 + c];
        const float v[4] = {sv.rawq[(size_t) gr * 4], sv.rawq[(size_t) gr * 4 + 1], sv.rawq[(size_t) gr * 4 + 2],
            sv.rawq[(size_t) gr * 4 + 3] + 1.0f};
        const float nv = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        if (nv > 1e-12f) {
          const float u[4] = {v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv};
          const float dot  = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2] + u[3] * gu[3];
#pragma unroll
          for (int c = 0; c < 4; ++c) g[3 + c] = (gu[c] - u[c] * dot) / nv;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) g[3 + c] = gu[c] / 1e-12f;
        }
        if (g_d_scale)
#pragma unroll
          for (int c = 0; c < 3; ++c) g[7 + c] = g_d_scale[(size_t) gr * 3 + c];
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      s_gh[tid][c]                     = g[c];
      wk.GH[(size_t) (r0 + tid) * 16 + c] = g[c];
    }
  }
  __syncthreads();
  // ---- gY_7 = gH W_heads: K = 10 (three 4-steps), tile c = features {64 wave + 4 j + c}
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int o   = 4 * s + q;
    const float a = s_gh[j][o];  // A[i = j][kk = q]
    float4 b      = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < NOUT) {
      int hrow;
      const int hd = head_of(o, hrow);
      b = *reinterpret_cast<const float4*>(n.head_w[hd] + (size_t) hrow * SPW + 64 * wave + 4 * j);
    }
    acc[0] = mfma4(a, b.x, acc[0]);
    acc[1] = mfma4(a, b.y, acc[1]);
    acc[2] = mfma4(a, b.z, acc[2]);
    acc[3] = mfma4(a, b.w, acc[3]);
  }
  int cur = 0;
  for (int l = SPD - 1; l >= 0; --l) {
    // gZ_l = gY_l * (Y_l > 0): lane holds rows 4 q + r, features 64 wave + 4 j + {0..3}
    const float* Yl = sv.Y + ((size_t) l * Mp + r0) * SPW + 64 * wave + 4 * j;
    float* GZl      = wk.GZ + ((size_t) l * Mp + r0) * SPW + 64 * wave + 4 * j;
    float* sz       = s_gz[cur] + 64 * wave + 4 * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row  = 4 * q + r;
      const float4 y = *reinterpret_cast<const float4*>(Yl + (size_t) row * SPW);
      const bool live = r0 + row < M;  // rows beyond M carry no gradient (their activations are copies of the last row's)
      float4 g;
      g.x = (live && y.x > 0.f) ? acc[0][r] : 0.f;
      g.y = (live && y.y > 0.f) ? acc[1][r] : 0.f;
      g.z = (live && y.z > 0.f) ? acc[2][r] : 0.f;
      g.w = (live && y.w > 0.f) ? acc[3][r] : 0.f;
      *reinterpret_cast<float4*>(sz + row * PITCH)           = g;
      *reinterpret_cast<float4*>(GZl + (size_t) row * SPW) = g;
    }
    if (l == 0) break;  // no gradient to the encoded input (the superpoint positions are detached)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_bwd(acc, s_gz[cur], PITCH, n.W[l], layer_ld(l), layer_hofs(l), wave, lane);
    cur ^= 1;
  }
}

// ================================================================================================ backward, launch B
// job table: [0,112) hidden x hidden products of layers 1..7 (16 tiles of 64 x 64 each; layer 5 writes at column 93),
// [112,120) layer 0 (256 x 93: 4 x 2 tiles), [120,128) layer 5's input part (256 x 93), [128,132) heads (10 x 256: 4 tiles of
// 16 x 64).  Rows (the contraction) split over the 4 waves, partial tiles summed through LDS.
constexpr int JOBS_HH = 7 * 16, JOBS_X0 = 8, JOBS_HEAD = 4, N_JOBS = JOBS_HH + 2 * JOBS_X0 + JOBS_HEAD;

__global__ void __launch_bounds__(NT) sp_net_backward_weights_kernel(int M, NetPtrs n, GradPtrs g, SavedView sv, WorkView wk) {
  __shared__ __attribute__((aligned(16))) float s_part[3][64 * 65];
  __shared__ float s_gb[4][64];
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int Mp  = pad_rows(M);
  const int job = blockIdx.x;
  // ---- decode
  int layer, o0, k0, kvalid, xld, gofs, gld;
  const float* X;     // right operand rows: [Mp][xld], columns k0 ...
  const float* A;     // left operand rows : GZ_l [Mp][256] or GH [Mp][16]
  float* G;           // output matrix
  bool heads = false, bias = false;
  if (job < JOBS_HH) {
    const int li = job / 16, t = job % 16;
    layer = li + 1, o0 = 64 * (t / 4), k0 = 64 * (t % 4), kvalid = SPW;
    X = sv.Y + (size_t) (layer - 1) * Mp * SPW, xld = SPW;
    A = wk.GZ + (size_t) layer * Mp * SPW;
    G = g.W[layer], gld = layer_ld(layer), gofs = layer_hofs(layer);
    bias = (t % 4) == 0;
  } else if (job < JOBS_HH + 2 * JOBS_X0) {
    const int t = (job - JOBS_HH) % JOBS_X0;
    layer = job < JOBS_HH + JOBS_X0 ? 0 : SKIP + 1;
    o0 = 64 * (t / 2), k0 = 64 * (t % 2), kvalid = IN0;
    X = sv.x0, xld = IN0P;
    A = wk.GZ + (size_t) layer * Mp * SPW;
    G = g.W[layer], gld = layer_ld(layer), gofs = 0;
    bias = layer == 0 && (t % 2) == 0;
  } else {
    heads = true, layer = SPD, o0 = 0, k0 = 64 * (job - JOBS_HH - 2 * JOBS_X0), kvalid = SPW;
    X = sv.Y + (size_t) (SPD - 1) * Mp * SPW, xld = SPW;
    A = wk.GH;
    G = nullptr, gld = SPW, gofs = 0;
    bias = k0 == 0;
  }
  // ---- the wave's share of the rows: rows [rb, re), 4 per step
  const int per = (Mp / 4 + 3) / 4 * 4;  // Mp is a multiple of 16: per = Mp / 4
  const int rb = wave * per, re = min(rb + per, Mp);
  f32x4 acc[4][4];  // [o tile a][k tile c]: outputs o0 + 4 i + a (heads: o = i), k0 + 4 j + c
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int kcol = k0 + 4 * j;
  if (!heads) {
    for (int r = rb; r < re; r += 4) {
      const float4 av = *reinterpret_cast<const float4*>(A + (size_t) (r + q) * SPW + o0 + 4 * j);   // lane (i = j, kk = q)
      const float4 bv = kcol < xld ? *reinterpret_cast<const float4*>(X + (size_t) (r + q) * xld + kcol) : make_float4(0, 0, 0, 0);
      colsum.x += av.x, colsum.y += av.y, colsum.z += av.z, colsum.w += av.w;
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = mfma4(a4[a], bv.x, acc[a][0]);
        acc[a][1] = mfma4(a4[a], bv.y, acc[a][1]);
        acc[a][2] = mfma4(a4[a], bv.z, acc[a][2]);
        acc[a][3] = mfma4(a4[a], bv.w, acc[a][3]);
      }
    }
  } else {
    for (int r = rb; r < re; r += 4) {
      const float av  = A[(size_t) (r + q) * 16 + j];  // GH row r + q, output column j (>= 10: zero)
      const float4 bv = *reinterpret_cast<const float4*>(X + (size_t) (r + q) * xld + kcol);
      colsum.x += av;
      acc[0][0] = mfma4(av, bv.x, acc[0][0]);
      acc[0][1] = mfma4(av, bv.y, acc[0][1]);
      acc[0][2] = mfma4(av, bv.z, acc[0][2]);
      acc[0][3] = mfma4(av, bv.w, acc[0][3]);
    }
  }
  // ---- waves 1..3 park their partial tiles in LDS ([o local (64)][k local (64)], pitch 65); wave 0 adds them to its own and
  // stores.  D layout: row 4 q + r of tile a <-> o local 4 (4 q + r) + a (heads: 4 q + r), column j of tile c <-> k local 4 j + c
  const int na = heads ? 1 : 4;
  {
    // column sums of the left operand over this wave's rows: lanes (j, q) hold features o0 + 4 j + {x,y,z,w} (heads: j)
    float4 cs = colsum;
    cs.x += __shfl_xor(cs.x, 16), cs.y += __shfl_xor(cs.y, 16), cs.z += __shfl_xor(cs.z, 16), cs.w += __shfl_xor(cs.w, 16);
    cs.x += __shfl_xor(cs.x, 32), cs.y += __shfl_xor(cs.y, 32), cs.z += __shfl_xor(cs.z, 32), cs.w += __shfl_xor(cs.w, 32);
    if (q == 0) {
      if (heads) {
        s_gb[wave][j] = cs.x;
      } else {
        s_gb[wave][4 * j] = cs.x, s_gb[wave][4 * j + 1] = cs.y, s_gb[wave][4 * j + 2] = cs.z, s_gb[wave][4 * j + 3] = cs.w;
      }
    }
  }
  if (wave > 0) {
    float* sp = s_part[wave - 1];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a >= na) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ol = heads ? 4 * q + r : 4 * (4 * q + r) + a;
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[ol * 65 + 4 * j + c] = acc[a][c][r];
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a >= na) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ol = heads ? 4 * q + r : 4 * (4 * q + r) + a;
        float* dst   = nullptr;
        if (heads) {
          if (ol < NOUT) {
            int hrow;
            const int hd = head_of(ol, hrow);
            dst = g.head_w[hd] + (size_t) hrow * SPW;
          }
        } else {
          dst = G + (size_t) (o0 + ol) * gld + gofs;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int kl  = 4 * j + c;
          const float v = (acc[a][c][r] + s_part[0][ol * 65 + kl]) + (s_part[1][ol * 65 + kl] + s_part[2][ol * 65 + kl]);
          if (dst && k0 + kl < kvalid) dst[k0 + kl] = v;
        }
      }
    }
  }
  if (bias && tid < (heads ? 16 : 64)) {
    const float v = (s_gb[0][tid] + s_gb[1][tid]) + (s_gb[2][tid] + s_gb[3][tid]);
    if (heads) {
      if (tid < NOUT) {
        int hrow;
        const int hd = head_of(tid, hrow);
        g.head_b[hd][hrow] = v;
      }
    } else {
      g.b[layer][o0 + tid] = v;
    }
  }
  // ---- the workgroup that finishes last: the time network's backward
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const unsigned t = atomicAdd(wk.ticket, 1u);
    s_last = (t == (unsigned) gridDim.x - 1) ? 1 : 0;
    if (s_last) *wk.ticket = 0u;  // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float* s_gt   = s_part[0];        // [32] d loss / d t_emb
  float* s_ghid = s_part[0] + 64;   // [256]
  if (tid < TOUT * 8) {
    const int c = tid >> 3, part = tid & 7;
    float v = 0.f;
    for (int o = part; o < SPW; o += 8)
      v += __builtin_nontemporal_load(&g.b[0][o]) * n.W[0][(size_t) o * IN0 + PDIM + c] +
           __builtin_nontemporal_load(&g.b[SKIP + 1][o]) * n.W[SKIP + 1][(size_t) o * (IN0 + SPW) + PDIM + c];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (part == 0) s_gt[c] = v;
  }
  __syncthreads();
  {  // second linear: gW2 [30][256] = g_t (x) hid, gb2 = g_t;  g_hid = W2^T g_t * (hid > 0)
    const float hid = sv.thid[tid];
    float gh = 0.f;
    for (int c = 0; c < TOUT; ++c) {
      g.tw2[c * THID + tid] = s_gt[c] * hid;
      gh += s_gt[c] * n.tw2[c * THID + tid];
    }
    gh = hid > 0.f ? gh : 0.f;
    s_ghid[tid] = gh;
    if (tid < TOUT) g.tb2[tid] = s_gt[tid];
    // first linear: gW1 [256][13] = g_hid (x) freq(t), gb1 = g_hid
    g.tb1[tid] = gh;
#pragma unroll
    for (int k = 0; k < TDIM; ++k) g.tw1[tid * TDIM + k] = gh * sv.temb[k];
  }
}

NetPtrs net_ptrs(const skgs_sp_net* d) {
  NetPtrs n;
  n.points = d->points, n.time = d->time;
  n.tw1 = d->time_w1, n.tb1 = d->time_b1, n.tw2 = d->time_w2, n.tb2 = d->time_b2;
  for (int l = 0; l < SPD; ++l) n.W[l] = d->W[l], n.b[l] = d->b[l];
  n.head_w[0] = d->warp_w, n.head_b[0] = d->warp_b;
  n.head_w[1] = d->rotation_w, n.head_b[1] = d->rotation_b;
  n.head_w[2] = d->scaling_w, n.head_b[2] = d->scaling_b;
  return n;
}
bool net_complete(const skgs_sp_net* d) {
  bool ok = d->time_w1 && d->time_b1 && d->time_w2 && d->time_b2 && d->warp_w && d->warp_b && d->scaling_w && d->scaling_b &&
            d->rotation_w && d->rotation_b;
  for (int l = 0; l < SPD; ++l) ok = ok && d->W[l] && d->b[l];
  return ok;
}
int cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_sp_net_saved_bytes(int32_t M) { return M > 0 ? saved_floats(M) * 4 : 0; }
size_t skgs_sp_net_workspace_bytes(int32_t M) { return M > 0 ? work_bytes(M) : 0; }

int skgs_sp_net_forward(const skgs_sp_net* net, float* raw, float* bone_T, float* d_rot, float* d_scale, void* saved,
    size_t saved_bytes, skgs_stream_t stream) {
  SKGS_REQUIRE(net && net->M >= 0, "sp_net_forward: NULL descriptor or M < 0");
  if (net->M == 0) return 0;
  SKGS_REQUIRE(net->points && net->time && net_complete(net), "sp_net_forward: NULL points / time / parameter");
  SKGS_REQUIRE(saved && saved_bytes >= skgs_sp_net_saved_bytes(net->M), "sp_net_forward: saved buffer too small");
  SKGS_REQUIRE(raw || bone_T, "sp_net_forward: no output requested");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_SP_NET_FWD, s);
  hipLaunchKernelGGL(sp_net_forward_kernel, dim3(pad_rows(net->M) / ROWS), dim3(NT), 0, s, net->M, net_ptrs(net), raw, bone_T, d_rot,
      d_scale, saved_view(saved, net->M));
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_sp_net_backward(const skgs_sp_net* net, const skgs_sp_net* grads, const float* g_bone_T, const float* g_d_rot,
    const float* g_d_scale, const float* g_raw, const void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
    const skgs_adam_range* side, skgs_stream_t stream) {
  SKGS_REQUIRE(net && grads && net->M >= 0, "sp_net_backward: NULL descriptor or M < 0");
  if (net->M == 0) return 0;
  SKGS_REQUIRE(net_complete(net) && net_complete(grads), "sp_net_backward: NULL parameter or gradient pointer");
  SKGS_REQUIRE(g_raw || g_bone_T || g_d_rot || g_d_scale, "sp_net_backward: no cotangent given");
  SKGS_REQUIRE(saved && saved_bytes >= skgs_sp_net_saved_bytes(net->M), "sp_net_backward: saved buffer too small");
  SKGS_REQUIRE(workspace && workspace_bytes >= skgs_sp_net_workspace_bytes(net->M), "sp_net_backward: workspace too small");
  hipStream_t s = (hipStream_t) stream;
  const int M = net->M, nblk = pad_rows(M) / ROWS;
  SideAdam sd{};
  int n_side = 0;
  if (side && side->n_tensors > 0) {
    SKGS_REQUIRE(side->tensors && side->step_count && side->chunk_begin >= 0 && side->chunk_end >= side->chunk_begin,
        "sp_net_backward: bad side range");
    sd.tensors = reinterpret_cast<const AdamTensor*>(side->tensors), sd.n = side->n_tensors;
    sd.c0 = side->chunk_begin, sd.c1 = side->chunk_end;
    sd.beta1 = side->beta1, sd.beta2 = side->beta2, sd.eps = (float) side->eps;
    sd.state = reinterpret_cast<const AdamState*>(side->step_count), sd.after_advance = side->after_advance ? 1 : 0;
    if (sd.c1 > sd.c0)  // two workgroups per idle CU (256 threads each): the stream needs the bytes in flight
      n_side = (int) std::max<long long>(1, std::min<long long>((sd.c1 - sd.c0 + 1) / 2, 2LL * std::max(cu_count() - nblk, 1)));
  }
  SavedView sv = saved_view(const_cast<void*>(saved), M);
  WorkView wk  = work_view(workspace, M);
  NetPtrs n    = net_ptrs(net);
  GradPtrs g;
  g.tw1 = const_cast<float*>(grads->time_w1), g.tb1 = const_cast<float*>(grads->time_b1);
  g.tw2 = const_cast<float*>(grads->time_w2), g.tb2 = const_cast<float*>(grads->time_b2);
  for (int l = 0; l < SPD; ++l) g.W[l] = const_cast<float*>(grads->W[l]), g.b[l] = const_cast<float*>(grads->b[l]);
  g.head_w[0] = const_cast<float*>(grads->warp_w), g.head_b[0] = const_cast<float*>(grads->warp_b);
  g.head_w[1] = const_cast<float*>(grads->rotation_w), g.head_b[1] = const_cast<float*>(grads->rotation_b);
  g.head_w[2] = const_cast<float*>(grads->scaling_w), g.head_b[2] = const_cast<float*>(grads->scaling_b);
  ProfScope prof(K_SP_NET_BWD, s);
  hipLaunchKernelGGL(sp_net_backward_rows_kernel, dim3(nblk + n_side), dim3(NT), 0, s, M, nblk, n, g_bone_T, g_d_rot, g_d_scale,
      g_raw, sv, wk, sd);
  SKGS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sp_net_backward_weights_kernel, dim3(N_JOBS), dim3(NT), 0, s, M, n, g, sv, wk);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
