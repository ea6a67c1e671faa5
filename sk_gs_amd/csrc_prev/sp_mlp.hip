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
//   forward    The ROWS are independent, so a workgroup takes FOUR superpoints through the WHOLE network: 128 workgroups, no
//              exchange between them, no barrier across the grid.  The weights stream from L2 (2 MB for all layers: every
//              XCD's 4 MB L2 holds them after the first touch) as rows of a TRANSPOSED copy ([contraction index][256 outputs],
//              written by a small launch in front: nn.Linear stores [out][in]) straight into the B operands of
//              v_mfma_f32_4x4x1_16B_f32; the 8 waves split the contraction, their partial tiles meet in LDS (`stream_rows`).
//   backward A the same row blocks walk back: gZ = gY * (Y > 0), gX = gZ W -- here the contraction index IS the row of nn.Linear's
//              matrix, so the stream reads the parameters themselves.  Every layer's gZ is saved for launch B.  The launch
//              occupies M / 4 = 128 CUs; its other workgroups run an optimizer piece (skgs_adam_range, as skgs_skeleton_backward
//              does).
//   backward B all weight gradients gW_l = gZ_l^T X_l (K = the 512 rows) as 64 x 64 output tiles on the whole chip: 132
//              workgroups, the rows split over the 4 waves and summed through LDS; bias gradients = column sums of gZ_l; the
//              time network's backward in the workgroup that finishes last (d loss / d t_emb = gb_0 W_0[:, 63:93] + gb_5
//              W_5[:, 63:93]: t_emb is the same for every row, so its gradient needs only the bias gradients).
// No gradient w.r.t. the superpoint positions: every caller detaches them (sk_gs.py:746-748,845).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

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
constexpr int ROWS   = 16;                   // row padding of the saved activations (the weight-gradient launch walks 16-row steps)
constexpr int RB     = 4;                    // superpoints per workgroup of the row-block launches = one 4 x 4 MFMA row group
constexpr int NT     = 512;                  // 8 waves: two per SIMD; wave w takes every 8th contraction row of every layer
constexpr int NTB    = 512;                  // the weight-gradient launch: 8 waves, the rows (the contraction) split over them
constexpr int NWB    = NTB / 64;
constexpr int NWAVE  = NT / 64;
constexpr int RING_F = 32, RING_B = 32;      // weight rows (1 KB each) a wave keeps in flight, forward / backward
constexpr int HT     = 36;                   // LDS pitch of a transposed activation row [r & 7][r >> 3]: 32 + 4
constexpr int XT     = 16;                   // ... of the encoded input: 12 + 4
constexpr int NOUT   = 10;                   // d_xyz 3 | d_rotation 4 | d_scaling 3
constexpr int NOUT_MAX = 14;                // ... | g_rotation 4 (sep_rot: the `local_rotation` head, sk_gs.py:275-282,315)

struct __attribute__((packed, aligned(4))) f4u {  // a float4 at 4-byte alignment (weight rows of 93 / 349 floats)
  float x, y, z, w;
};
__device__ __forceinline__ float4 ldg4(const float* p) {
  const f4u v = *reinterpret_cast<const f4u*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 outer products, D_b[i][j] += A_b[i] B_b[j]; lane 4 b + i holds A_b[i], lane
// 4 b + j holds B_b[j], register i of lane 4 b + j holds D_b[i][j] (checked on the chip: tools/micro/mfma4x4_layer.hip)
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// forward stream of a wave: contraction rows per layer / 8 (the encoded input is padded to 96 columns; layer 5 contracts
// [input 96 | hidden 256]) and where a layer starts in the flattened sequence
__host__ __device__ constexpr int fwd_items(int l) { return l == 0 ? IN0P / 8 : (l == SKIP + 1 ? (IN0P + SPW) / 8 : SPW / 8); }
__host__ __device__ constexpr int fwd_start(int l) {
  int t = 0;
  for (int i = 0; i < l; ++i) t += fwd_items(i);
  return t;
}
constexpr int FWD_TOTAL = fwd_start(SPD);   // 248 items per wave
constexpr int WT_ROWS   = 8 * FWD_TOTAL;    // 1984 rows of 256 floats (2.03 MB)
constexpr int BWD_TOTAL = (SPD - 1) * SPW / 8;  // 224: layers 7 .. 1, 32 rows each per wave

// saved by the forward for the backward (floats), Mp = rows rounded up to 16
struct SavedView {
  float* x0;    // [Mp][IN0P]   encoded input (columns 93..95 zero)
  float* Y;     // [SPD][Mp][SPW] post-ReLU activations
  float* rawq;  // [Mp][4]      raw rotation head (before + [0,0,0,1] and the normalisation)
  float* rawl;  // [Mp][4]      raw local-rotation head (sep_rot), likewise
  float* temb;  // [16]         freq(t) (13 used)
  float* thid;  // [THID]       hidden layer of the time network (post-ReLU)
  float* tout;  // [32]         its output (30 used)
  float* wt;    // [WT_ROWS][SPW] the hidden layers' weights with the CONTRACTION index as the row (written by every forward)
};
__host__ __device__ inline int pad_rows(int M) { return (M + ROWS - 1) / ROWS * ROWS; }
__host__ __device__ inline size_t saved_floats(int M) {
  const size_t Mp = pad_rows(M);
  return Mp * IN0P + (size_t) SPD * Mp * SPW + Mp * 8 + 16 + THID + 32 + (size_t) WT_ROWS * SPW;
}
__host__ __device__ inline SavedView saved_view(void* base, int M) {
  const size_t Mp = pad_rows(M);
  SavedView v;
  float* p = reinterpret_cast<float*>(base);
  v.x0 = p, p += Mp * IN0P;
  v.Y = p, p += (size_t) SPD * Mp * SPW;
  v.rawq = p, p += Mp * 4;
  v.rawl = p, p += Mp * 4;
  v.temb = p, p += 16;
  v.thid = p, p += THID;
  v.tout = p, p += 32;
  v.wt = p;
  return v;
}
// backward workspace: (256 B reserved) | GH [Mp][16] head cotangents (10 used) | GZ [SPD][Mp][SPW] | GBP [2][Mp / 4][SPW]: the
// row blocks' column sums of gZ_0 and gZ_5 (the time network's backward needs the bias gradients of those two layers)
struct WorkView {
  unsigned* ticket;
  float* GH;
  float* GZ;
  float* GBP;
};
__host__ __device__ inline size_t work_bytes(int M) {
  const size_t Mp = pad_rows(M);
  return 256 + (Mp * 16 + (size_t) SPD * Mp * SPW + 2 * (Mp / RB) * SPW) * 4;
}
__host__ __device__ inline WorkView work_view(void* base, int M) {
  const size_t Mp = pad_rows(M);
  WorkView v;
  v.ticket = reinterpret_cast<unsigned*>(base);
  v.GH     = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + 256);
  v.GZ     = v.GH + Mp * 16;
  v.GBP    = v.GZ + (size_t) SPD * Mp * SPW;
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
// (in0: 93 with the time network; 63 + 1 + 2 deg with the raw time encoding of is_blender = False, sk_gs.py:255-261)
__host__ __device__ inline int layer_ld(int l, int in0) { return l == 0 ? in0 : (l == SKIP + 1 ? in0 + SPW : SPW); }
__host__ __device__ inline int layer_hofs(int l, int in0) { return l == SKIP + 1 ? in0 : 0; }

// ---- the product of a row block with a weight matrix, streamed -----------------------------------------------------------
// Found with tools/micro/mfma4x4_layer.hip (MI355X).  A CU pulls weights out of L2 at most at its L1's 64 B/clk (~1.7 us for a
// 256 x 256 fp32 layer), and only with wave instructions whose consecutive lanes read consecutive addresses; the fp32 matrix
// pipe does 128 multiply-adds per clock and CU.  Round 4 first ran 16-row blocks on v_mfma_f32_16x16x4_f32: 32 workgroups,
// 3.4 us of MFMA issue per layer each and the operand re-shaped through LDS -- 5.7 us per layer.  Here a workgroup takes FOUR
// rows (128 workgroups for 512 superpoints) on v_mfma_f32_4x4x1_16B_f32 with the weights stored with the CONTRACTION index as
// the row ([r][256 outputs]: nn.Linear's own layout for the backward, a transposed copy for the forward):
//   * lane l loads outputs 4 l .. 4 l + 3 of row r -- one row x 1 KB per wave instruction, straight into the B operands of four
//     MFMAs (register q: block b, column j <-> output 16 b + 4 j + q); the A operand x[i][r] is the same in all 16 blocks;
//   * the 8 waves split the contraction: wave w takes rows r = 8 t + w.  32 rows per wave (256 KB per CU: a whole
//     layer) are in flight at any time, each register refilled with the row 32 steps ahead right after its MFMAs -- across
//     layer boundaries, the sequence of a wave is one flat list (everything below is unrolled: the waits are exact vmcnt);
//   * the waves' partial [4 x 256] tiles meet in LDS (32 KB), two barriers per layer.
// 2.5 us per layer instead of 5.7 (the micro: 4 rows / 8 waves; 8 rows per workgroup 3.5, no split 2.6).
// Activations live in LDS transposed, [row][r & 7][r >> 3], so that the A operands of four steps are one ds_read_b128.
template <int T0, int NR, int TOTAL, int RING, class RowPtr>
__device__ __forceinline__ void stream_rows(f32x4 (&acc)[4], float4 (&ring)[RING], const float* ap, const RowPtr& row_ptr) {
  static_assert(NR % 4 == 0, "four steps per operand read");
  float4 a = *reinterpret_cast<const float4*>(ap);
#pragma unroll
  for (int t4 = 0; t4 < NR / 4; ++t4) {
    const float4 ac = a;
    if (t4 + 1 < NR / 4) a = *reinterpret_cast<const float4*>(ap + 4 * (t4 + 1));
    const float av[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = T0 + 4 * t4 + e, slot = t % RING;
      acc[0] = mfma1(av[e], ring[slot].x, acc[0]);
      acc[1] = mfma1(av[e], ring[slot].y, acc[1]);
      acc[2] = mfma1(av[e], ring[slot].z, acc[2]);
      acc[3] = mfma1(av[e], ring[slot].w, acc[3]);
      __builtin_amdgcn_sched_barrier(0);  // (left alone the compiler sinks every refill to the end of the layer: load phase,
      if (t + RING < TOTAL) ring[slot] = ldg4(row_ptr(t + RING));  // then MFMA phase, 3.3 us per layer)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the wave's partial tile -> LDS.  Register i of acc[q] in lane l = row i, output 4 l + q
__device__ __forceinline__ void park_partial(const f32x4 (&acc)[4], float (*part)[SPW], int lane) {
#pragma unroll
  for (int i = 0; i < RB; ++i)
    *reinterpret_cast<float4*>(&part[i][4 * lane]) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
}
// thread (i = tid >> 7, o = 2 (tid & 127)): the 8 partials of two neighbouring outputs
__device__ __forceinline__ float2 sum_partials(const float (*part)[RB][SPW], int i, int o, float2 v) {
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) {
    const float2 p = *reinterpret_cast<const float2*>(&part[w][i][o]);
    v.x += p.x, v.y += p.y;
  }
  return v;
}
__device__ __forceinline__ void zero4(f32x4 (&acc)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

struct NetPtrs {  // device copy of skgs_sp_net's pointers
  const float* points;
  const float* time;
  const float *tw1, *tb1, *tw2, *tb2;
  const float* W[SPD];
  const float* b[SPD];
  const float* head_w[4];  // warp (3), rotation (4), scaling (3), local rotation (4, sep_rot only): the order of the raw output row
  const float* head_b[4];
  int nout;                // 10, or 14 with the local-rotation head
  int lbs_c;               // warp_method LBS_c: bone_T's translation is d_xyz + x + R(u)(-x) (sk_gs.py:803-804)
  int in0;                 // columns of the encoded input: 93 = 63 + 30 (time network), or 63 + tdim (raw time encoding)
  int tdim;                // raw time encoding (is_blender = False: no time network, tw1 == NULL): 1 + 2 * degree columns
};
struct GradPtrs {
  float *tw1, *tb1, *tw2, *tb2;
  float* W[SPD];
  float* b[SPD];
  float* head_w[4];
  float* head_b[4];
  float* points;  // LBS_c: d loss / d sp_points [M,3] (written), or NULL
};
__device__ __forceinline__ int head_of(int o, int& row) {  // raw output column -> (head, row of that head's matrix)
  if (o < 3) return row = o, 0;
  if (o < 7) return row = o - 3, 1;
  if (o < 10) return row = o - 7, 2;
  return row = o - 10, 3;
}
// y = R(q) v for a unit quaternion q = (a, w): v + 2 w (a x v) + 2 a x (a x v)   (lie.h:59-64)
__device__ __forceinline__ void quat_rotate(const float* q, const float* v, float* y) {
  float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
  y[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  y[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  y[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
// d (g . R(q) v) / d q for the four stored numbers of a unit q (the normalisation's Jacobian is applied by the caller):
//   d/dw = 2 g . (a x v);   d/da = 2 w (v x g) + 2 [g (a . v) + v (g . a) - 2 a (g . v)]
__device__ __forceinline__ void quat_rotate_grad_q(const float* q, const float* v, const float* g, float* gq) {
  const float a[3] = {q[0], q[1], q[2]}, w = q[3];
  const float axv[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};
  const float vxg[3] = {v[1] * g[2] - v[2] * g[1], v[2] * g[0] - v[0] * g[2], v[0] * g[1] - v[1] * g[0]};
  const float av = a[0] * v[0] + a[1] * v[1] + a[2] * v[2], ga = g[0] * a[0] + g[1] * a[1] + g[2] * a[2];
  const float gv = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) gq[c] = 2.f * w * vxg[c] + 2.f * (g[c] * av + v[c] * ga - 2.f * a[c] * gv);
  gq[3] = 2.f * (g[0] * axv[0] + g[1] * axv[1] + g[2] * axv[2]);
}

// =================================================================================================== forward
// The hidden layers' weights with the contraction index as the row, for the forward's stream: layer 0 [96][256] (rows 93..95
// zero), layers 1-4, 6, 7 [256][256], layer 5 [96 | 256][256] (its input part padded like layer 0, then the hidden part) --
// WT_ROWS = 1984 rows.  32 x 32 tiles through LDS; a tile never straddles two layers (96, 256, 352 are multiples of 32).
constexpr int TRANSPOSE_GROUPS = WT_ROWS / 32 * (SPW / 32);
__global__ void __launch_bounds__(256) sp_net_transpose_kernel(NetPtrs n, float* __restrict__ wt, SpPrepareJob prep) {
  if ((int) blockIdx.x >= TRANSPOSE_GROUPS) {  // the search's table and list counters for this step (skgs_sp_prepare)
    sp_prepare_element(prep, ((int) blockIdx.x - TRANSPOSE_GROUPS) * 256 + (int) threadIdx.x);
    return;
  }
  __shared__ float tile[32][33];
  const int rb = blockIdx.x >> 3, nb = blockIdx.x & 7, R0 = 32 * rb;
  int l = 0, base = 0;
  for (; l < SPD - 1; ++l) {
    if (R0 < base + 8 * fwd_items(l)) break;
    base += 8 * fwd_items(l);
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, rl = R0 - base + tx;  // rl: the row inside the layer
  int k = rl;
  bool valid = true;
  if (l == 0) valid = rl < n.in0;
  if (l == SKIP + 1) {
    if (rl < IN0P) valid = rl < n.in0;
    else k = n.in0 + rl - IN0P;
  }
  const float* W = n.W[l];
  const int ld   = layer_ld(l, n.in0);
#pragma unroll
  for (int yy = ty; yy < 32; yy += 8) tile[yy][tx] = valid ? W[(size_t) (32 * nb + yy) * ld + k] : 0.f;
  __syncthreads();
#pragma unroll
  for (int yy = ty; yy < 32; yy += 8) wt[(size_t) (R0 + yy) * SPW + 32 * nb + tx] = tile[tx][yy];
}

struct FwdRows {  // item t of wave w = row 8 t + w of the transposed weights
  const float* wt;
  int at;  // wave * SPW + 4 * lane
  __device__ __forceinline__ const float* operator()(int t) const { return wt + (uint32_t) (t * (8 * SPW) + at); }
};

__global__ void __launch_bounds__(NT) sp_net_forward_kernel(int M, NetPtrs n, float* __restrict__ raw, float* __restrict__ bone_T,
    float* __restrict__ d_rot, float* __restrict__ d_scale, SavedView sv) {
  __shared__ __attribute__((aligned(16))) float s_x0t[RB][8][XT];     // encoded input, transposed: [row][c & 7][c >> 3]
  __shared__ __attribute__((aligned(16))) float s_ht[2][RB][8][HT];   // activations, two buffers, transposed likewise
  __shared__ __attribute__((aligned(16))) float s_part[NWAVE][RB][SPW];
  __shared__ float s_temb[32], s_thid[THID], s_tout[32];
  __shared__ float s_raw[RB][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i_ = lane & 3;
  const int r0 = blockIdx.x * RB, Mp = pad_rows(M);
  // ---- the first 32 weight rows of this wave are on their way while the time network runs
  const int ei = tid >> 7, eo = 2 * (tid & 127);  // the thread's share of a layer's epilogue: row ei, outputs eo, eo + 1
  float2 bias[SPD];  // (loaded FIRST: vmcnt counts in order -- a load issued at a layer's end would wait for the whole ring)
#pragma unroll
  for (int l = 0; l < SPD; ++l) bias[l] = *reinterpret_cast<const float2*>(n.b[l] + eo);
  const FwdRows rows{sv.wt, wave * SPW + 4 * lane};
  float4 ring[RING_F];
#pragma unroll
  for (int t = 0; t < RING_F; ++t) ring[t] = ldg4(rows(t));
  // ---- time network (every workgroup: 13 -> 256 -> 30 is ~11 k multiply-adds)
  const bool timenet = n.tw1 != nullptr;  // (uniform)
  if (tid < (timenet ? TDIM : n.tdim)) {
    const float t = n.time[0];
    float v = t;
    if (tid >= 1) {
      const int col = tid - 1;
      v = sinf(scalbnf(t, col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
    }
    s_temb[tid] = v;
    if (!timenet) s_tout[tid] = v;  // is_blender = False: t_emb IS freq(t) (sk_gs.py:297-299 without the timenet line)
  }
  __syncthreads();
  if (timenet) {
  if (tid < THID) {
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
  }
  if (blockIdx.x == 0 && timenet) {
    if (tid < 16) sv.temb[tid] = tid < TDIM ? s_temb[tid] : 0.f;
    if (tid < THID) sv.thid[tid] = s_thid[tid];
    if (tid < 32) sv.tout[tid] = tid < TOUT ? s_tout[tid] : 0.f;
  }
  // ---- encoded input of the 4 rows: [x | sin / cos(2^f x) ...] (freqencoder.cu:7-31) | t_emb | 0 0 0
  if (tid < RB * IN0P) {
    const int row = tid / IN0P, c = tid - row * IN0P;
    const int gr = min(r0 + row, M - 1);  // (rows beyond M repeat the last one: computed, never stored outside `saved`)
    float v = 0.f;
    if (c < 3) {
      v = n.points[3 * gr + c];
    } else if (c < PDIM) {
      const int col = c / 3 - 1, d = c % 3;
      v = sinf(scalbnf(n.points[3 * gr + d], col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
    } else if (c < n.in0) {
      v = s_tout[c - PDIM];
    }
    s_x0t[row][c & 7][c >> 3]              = v;
    sv.x0[(size_t) (r0 + row) * IN0P + c] = v;
  }
  __syncthreads();
  // ---- the eight layers: one flat stream of FWD_TOTAL items per wave
  f32x4 acc[4];
  int cur = 0;
  auto close_layer = [&](int l) {  // partial tiles -> bias, ReLU -> the next layer's operand (LDS, transposed) and `saved`
    park_partial(acc, s_part[wave], lane);
    __syncthreads();
    float2 v = sum_partials(s_part, ei, eo, bias[l]);
    v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f);
    *reinterpret_cast<float2*>(sv.Y + ((size_t) l * Mp + r0 + ei) * SPW + eo) = v;
    s_ht[cur ^ 1][ei][eo & 7][eo >> 3]       = v.x;
    s_ht[cur ^ 1][ei][(eo & 7) + 1][eo >> 3] = v.y;
    cur ^= 1;
    __syncthreads();
  };
  const float* ax = &s_x0t[i_][wave][0];
#define SP_HID(T0)  stream_rows<T0, SPW / 8, FWD_TOTAL, RING_F>(acc, ring, &s_ht[cur][i_][wave][0], rows)
#define SP_X0(T0)   stream_rows<T0, IN0P / 8, FWD_TOTAL, RING_F>(acc, ring, ax, rows)
  zero4(acc), SP_X0(fwd_start(0)), close_layer(0);
  zero4(acc), SP_HID(fwd_start(1)), close_layer(1);
  zero4(acc), SP_HID(fwd_start(2)), close_layer(2);
  zero4(acc), SP_HID(fwd_start(3)), close_layer(3);
  zero4(acc), SP_HID(fwd_start(4)), close_layer(4);
  zero4(acc), SP_X0(fwd_start(5)), SP_HID(fwd_start(5) + IN0P / 8), close_layer(5);
  zero4(acc), SP_HID(fwd_start(6)), close_layer(6);
  zero4(acc), SP_HID(fwd_start(7)), close_layer(7);
#undef SP_HID
#undef SP_X0
  // ---- heads: raw[4 x nout] = h W_heads^T + b.  Thread (row i, output o, part p): the contraction indices k = 8 kk + p
  {
    const int pp = tid & 7, o = (tid >> 3) & 15, i = tid >> 7;
    int hrow;
    const int hd    = head_of(min(o, n.nout - 1), hrow);
    const float* wr = n.head_w[hd] + (size_t) hrow * SPW + pp;
    const float* hp = &s_ht[cur][i][pp][0];
    float w[SPW / 8];
#pragma unroll
    for (int kk = 0; kk < SPW / 8; ++kk) w[kk] = wr[8 * kk];  // (one round of loads)
    float v = 0.f;
#pragma unroll
    for (int kk = 0; kk < SPW / 8; ++kk) v += hp[kk] * w[kk];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (pp == 0 && o < n.nout) s_raw[i][o] = v + n.head_b[hd][hrow];
  }
  __syncthreads();
  if (tid < RB) {  // one thread per superpoint: the three raw outputs and the stage's epilogue (sk_gs.py:847)
    const int gr = r0 + tid;
    float o[NOUT_MAX];
#pragma unroll
    for (int c = 0; c < NOUT_MAX; ++c) o[c] = c < n.nout ? s_raw[tid][c] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) sv.rawq[(size_t) gr * 4 + c] = o[3 + c], sv.rawl[(size_t) gr * 4 + c] = o[10 + c];
    if (gr < M) {
      if (raw)
#pragma unroll
        for (int c = 0; c < NOUT_MAX; ++c)
          if (c < n.nout) raw[(size_t) gr * n.nout + c] = o[c];
      const float v[4] = {o[3], o[4], o[5], o[6] + 1.0f};
      const float nv   = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]), 1e-12f);  // F.normalize eps
      const float u[4] = {v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv};
      if (bone_T) {
        float t[3] = {o[0], o[1], o[2]};
        if (n.lbs_c) {  // sp_t = d_xyz + sp_points + SO3(d_rot).act(-sp_points)   (sk_gs.py:803-804)
          const float x[3] = {n.points[(size_t) gr * 3], n.points[(size_t) gr * 3 + 1], n.points[(size_t) gr * 3 + 2]};
          const float mx[3] = {-x[0], -x[1], -x[2]};
          float y[3];
          quat_rotate(u, mx, y);
#pragma unroll
          for (int c = 0; c < 3; ++c) t[c] = (o[c] + x[c]) + y[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) bone_T[(size_t) gr * 7 + c] = t[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) bone_T[(size_t) gr * 7 + 3 + c] = u[c];
      }
      if (d_rot) {  // what warp blends: the unit d_rot, or with sep_rot the unit g_rot (sk_gs.py:848,818-821)
        if (n.nout == NOUT_MAX) {
          const float l[4] = {o[10], o[11], o[12], o[13] + 1.0f};
          const float nl   = fmaxf(sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2] + l[3] * l[3]), 1e-12f);
#pragma unroll
          for (int c = 0; c < 4; ++c) d_rot[(size_t) gr * 4 + c] = l[c] / nl;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) d_rot[(size_t) gr * 4 + c] = u[c];
        }
      }
      if (d_scale)
#pragma unroll
        for (int c = 0; c < 3; ++c) d_scale[(size_t) gr * 3 + c] = o[7 + c];
    }
  }
}

// ================================================================================================ backward, launch A
__device__ __forceinline__ void side_adam_walk(const SideAdam& a, int wg, int n_side) {
  const int half = threadIdx.x >> 8, t256 = threadIdx.x & 255, lane = threadIdx.x & 63;
  const AdamCoef k            = adam_coefficients(a.beta1, a.beta2, a.eps, a.state, a.after_advance != 0);
  const AdamTensorLanes desc  = adam_load_descriptors(a.tensors, a.n, lane);
  const int64_t first0        = lane < a.n ? a.tensors[lane].chunk0 : INT64_MAX;
  const int64_t n_chunks      = a.c1 - a.c0;
  const int64_t begin = a.c0 + n_chunks * wg / n_side, end = a.c0 + n_chunks * (wg + 1) / n_side;
#if defined(SKGS_SP_SIDE_CHUNKS) && SKGS_SP_SIDE_CHUNKS == 1   // (A/B: one chunk per half and iteration pays beside the 20-row network,
                            // mlp_fused.hip::adam_side_job, not here: 0.3990 -> 0.4010 ms per step)
  for (int64_t chunk = begin + half; chunk < end; chunk += 2) {
    const int ti0      = adam_owner(a.tensors, a.n, first0, lane, chunk);
    const AdamTensor T0 = ti0 < 64 ? adam_descriptor_of(desc, ti0) : a.tensors[ti0];
    adam_update_chunk(T0, (chunk - T0.chunk0) * ADAM_CHUNK, t256, k);
  }
  return;
#endif
  for (int64_t chunk = begin + 2 * half; chunk < end; chunk += 4) {
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

struct BwdRows {  // item t of wave w: layer 7 - t / 32, row (output feature) 8 (t % 32) + w of that layer's weight matrix,
  const float* W[SPD];  // the 256 columns of its hidden part
  int wave, lane, in0;
  __device__ __forceinline__ const float* operator()(int t) const {
    const int l = SPD - 1 - t / (SPW / 8), o = 8 * (t % (SPW / 8)) + wave;
    return W[l] + (uint32_t) (o * layer_ld(l, in0) + layer_hofs(l, in0) + 4 * lane);  // (uniform base + 32-bit lane offset: one VGPR per address)
  }
};

__global__ void __launch_bounds__(NT) sp_net_backward_rows_kernel(int M, int nblk, NetPtrs n, const float* __restrict__ g_bone_T,
    const float* __restrict__ g_d_rot, const float* __restrict__ g_d_scale, const float* __restrict__ g_raw,
    float* __restrict__ g_points, SavedView sv, WorkView wk, SideAdam side) {
  if ((int) blockIdx.x >= nblk) {  // the CUs the row blocks leave idle: an optimizer piece
    side_adam_walk(side, (int) blockIdx.x - nblk, (int) gridDim.x - nblk);
    return;
  }
  __shared__ __attribute__((aligned(16))) float s_gzt[2][RB][8][HT];  // gZ of the current layer, transposed [row][o & 7][o >> 3]
  __shared__ __attribute__((aligned(16))) float s_part[NWAVE][RB][SPW];
  __shared__ float s_gh[RB][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i_ = lane & 3;
  const int r0 = blockIdx.x * RB, Mp = pad_rows(M);
  BwdRows rows;
#pragma unroll
  for (int l = 0; l < SPD; ++l) rows.W[l] = n.W[l];
  rows.wave = wave, rows.lane = lane, rows.in0 = n.in0;
  float4 ring[RING_B];
#pragma unroll
  for (int t = 0; t < RING_B; ++t) ring[t] = ldg4(rows(t));
  // ---- cotangent of the raw output row [d_xyz 3 | rotation 4 | scaling 3]
  if (tid < RB) {
    const int gr = r0 + tid;
    float g[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) g[c] = 0.f;
    if (gr < M) {
      if (g_raw) {
#pragma unroll
        for (int c = 0; c < NOUT_MAX; ++c)
          if (c < n.nout) g[c] = g_raw[(size_t) gr * n.nout + c];
      } else {
        // bone_T = [t | u], u = v / |v|, v = rotation + [0,0,0,1]:  g_v = (g_u - u (u . g_u)) / |v|;  t = d_xyz, or with LBS_c
        // d_xyz + x + R(u)(-x).  d_rot (the blended rotation) = u, or with the local-rotation head normalize(local + [0,0,0,1])
        const bool sep = n.nout == NOUT_MAX;
        float gu[4] = {0.f, 0.f, 0.f, 0.f}, gl[4] = {0.f, 0.f, 0.f, 0.f};
        const float v[4] = {sv.rawq[(size_t) gr * 4], sv.rawq[(size_t) gr * 4 + 1], sv.rawq[(size_t) gr * 4 + 2],
            sv.rawq[(size_t) gr * 4 + 3] + 1.0f};
        const float nv = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        const float nvc = fmaxf(nv, 1e-12f);
        const float u[4] = {v[0] / nvc, v[1] / nvc, v[2] / nvc, v[3] / nvc};
        if (g_bone_T) {
#pragma unroll
          for (int c = 0; c < 3; ++c) g[c] = g_bone_T[(size_t) gr * 7 + c];
#pragma unroll
          for (int c = 0; c < 4; ++c) gu[c] = g_bone_T[(size_t) gr * 7 + 3 + c];
          if (n.lbs_c) {  // t = d_xyz + x + R(u)(-x):  g_x = g_t - R(u)^T g_t,  g_u += d (g_t . R(u)(-x)) / d u
            const float x[3] = {n.points[(size_t) gr * 3], n.points[(size_t) gr * 3 + 1], n.points[(size_t) gr * 3 + 2]};
            const float mx[3] = {-x[0], -x[1], -x[2]}, gt[3] = {g[0], g[1], g[2]};
            float gq[4], rt[3];
            quat_rotate_grad_q(u, mx, gt, gq);
#pragma unroll
            for (int c = 0; c < 4; ++c) gu[c] += gq[c];
            if (g_points) {
              const float uc[4] = {-u[0], -u[1], -u[2], u[3]};
              quat_rotate(uc, gt, rt);
#pragma unroll
              for (int c = 0; c < 3; ++c) g_points[(size_t) gr * 3 + c] = gt[c] - rt[c];
            }
          }
        } else if (n.lbs_c && g_points) {
#pragma unroll
          for (int c = 0; c < 3; ++c) g_points[(size_t) gr * 3 + c] = 0.f;
        }
        if (g_d_rot)
#pragma unroll
          for (int c = 0; c < 4; ++c) (sep ? gl[c] : gu[c]) += g_d_rot[(size_t) gr * 4 + c];
        if (nv > 1e-12f) {
          const float dot  = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2] + u[3] * gu[3];
#pragma unroll
          for (int c = 0; c < 4; ++c) g[3 + c] = (gu[c] - u[c] * dot) / nv;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) g[3 + c] = gu[c] / 1e-12f;
        }
        if (sep) {
          const float l[4] = {sv.rawl[(size_t) gr * 4], sv.rawl[(size_t) gr * 4 + 1], sv.rawl[(size_t) gr * 4 + 2],
              sv.rawl[(size_t) gr * 4 + 3] + 1.0f};
          const float nl = sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2] + l[3] * l[3]);
          if (nl > 1e-12f) {
            const float ul[4] = {l[0] / nl, l[1] / nl, l[2] / nl, l[3] / nl};
            const float dot   = ul[0] * gl[0] + ul[1] * gl[1] + ul[2] * gl[2] + ul[3] * gl[3];
#pragma unroll
            for (int c = 0; c < 4; ++c) g[10 + c] = (gl[c] - ul[c] * dot) / nl;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) g[10 + c] = gl[c] / 1e-12f;
          }
        }
        if (g_d_scale)
#pragma unroll
          for (int c = 0; c < 3; ++c) g[7 + c] = g_d_scale[(size_t) gr * 3 + c];
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      s_gh[tid][c]                        = g[c];
      wk.GH[(size_t) (r0 + tid) * 16 + c] = g[c];
    }
  }
  __syncthreads();
  // ---- thread (row ei, features eo, eo + 1) closes every layer: gZ_l = gY_l * (Y_l > 0) -> `GZ` (for launch B), the next
  // product's operand (LDS, transposed), and this block's column sums of gZ_0 / gZ_5 (for the time network, launch B)
  const int ei = tid >> 7, eo = 2 * (tid & 127);
  const bool live = r0 + ei < M;  // rows beyond M carry no gradient (their activations are copies of the last row's)
  int cur = 0;
  auto y_of = [&](int l) { return *reinterpret_cast<const float2*>(sv.Y + ((size_t) l * Mp + r0 + ei) * SPW + eo); };
  auto close_layer = [&](int l, float2 gy, float2 y) {
    const size_t at = ((size_t) l * Mp + r0 + ei) * SPW + eo;
    float2 g;
    g.x = (live && y.x > 0.f) ? gy.x : 0.f;
    g.y = (live && y.y > 0.f) ? gy.y : 0.f;
    *reinterpret_cast<float2*>(wk.GZ + at) = g;
    s_gzt[cur][ei][eo & 7][eo >> 3]       = g.x;
    s_gzt[cur][ei][(eo & 7) + 1][eo >> 3] = g.y;
    __syncthreads();
    if ((l == 0 || l == SKIP + 1) && tid < SPW) {
      float cs = 0.f;
#pragma unroll
      for (int row = 0; row < RB; ++row) cs += s_gzt[cur][row][tid & 7][tid >> 3];
      wk.GBP[((size_t) (l == 0 ? 0 : 1) * nblk + blockIdx.x) * SPW + tid] = cs;
    }
  };
  {  // gY_7 = gH W_heads (K = 10, or 14 with the local-rotation head)
    float2 gy = make_float2(0.f, 0.f);
#pragma unroll
    for (int c = 0; c < NOUT_MAX; ++c) {
      if (c >= n.nout) break;
      int hrow;
      const int hd   = head_of(c, hrow);
      const float2 w = *reinterpret_cast<const float2*>(n.head_w[hd] + (size_t) hrow * SPW + eo);
      const float gc = s_gh[ei][c];
      gy.x += gc * w.x, gy.y += gc * w.y;
    }
    close_layer(SPD - 1, gy, y_of(SPD - 1));
  }
  // ---- gY_{l-1} = gZ_l W_l[:, hidden part], l = 7 .. 1: one flat stream of BWD_TOTAL items per wave.  (No gradient to the
  // encoded input: the superpoint positions are detached.)
  f32x4 acc[4];
  float2 y_in;  // Y_{l-1} of this thread's two features, requested BEFORE the stream of layer l (vmcnt counts in order: a load
                // issued at the layer's end would wait for the whole ring)
#define SP_BACK(L)                                                                                              \
  y_in = y_of((L) - 1);                                                                                         \
  zero4(acc);                                                                                                   \
  stream_rows<(SPD - 1 - (L)) * (SPW / 8), SPW / 8, BWD_TOTAL, RING_B>(acc, ring, &s_gzt[cur][i_][wave][0], rows);      \
  park_partial(acc, s_part[wave], lane);                                                                        \
  __syncthreads();                                                                                              \
  cur ^= 1;                                                                                                     \
  close_layer((L) - 1, sum_partials(s_part, ei, eo, make_float2(0.f, 0.f)), y_in);
  SP_BACK(7) SP_BACK(6) SP_BACK(5) SP_BACK(4) SP_BACK(3) SP_BACK(2) SP_BACK(1)
#undef SP_BACK
}

// The time network's backward (one workgroup of launch B).  t_emb is the same for every row, so
//   d loss / d t_emb [30] = gb_0 W_0[:, 63:93] + gb_5 W_5[:, 63:93]
// needs only the bias gradients of layers 0 and 5 -- summed here from the row blocks' partial column sums (launch A), so the
// job depends on no other workgroup of its launch (a last-workgroup-out ticket needed a device-scope fence per workgroup:
// an L2 write-back each, 30 us of the launch).
__device__ void timenet_backward(int M, const NetPtrs& n, const GradPtrs& g, const SavedView& sv, const WorkView& wk, float* s_buf) {
  const int tid = threadIdx.x, Mp = pad_rows(M);
  // d loss / d t_emb [30] = gb_0 W_0[:, 63:93] + gb_5 W_5[:, 63:93]: thread o forms its 30 products from ONE round of loads
  // (a loop over o per output was 32 dependent round trips: 40 us of this launch), parked [c][o] in LDS, 8 lanes sum a column
  float* s_prod = s_buf;              // [TOUT][256]  (30 KB of the 49 KB partial-tile area)
  float* s_gt   = s_buf + TOUT * SPW; // [32]
  float* s_ghid = s_gt + 64;              // [256]
  const bool act = tid < SPW;             // (the launch has 512 threads; one per feature works here, all reach the barriers)
  if (act) {
    const int nblk = Mp / RB;
    float gb0 = 0.f, gb5 = 0.f;  // bias gradients of layers 0 and 5, from the row blocks' partial sums (launch A)
    for (int b0 = 0; b0 < nblk; b0 += 16) {  // 32 loads in flight per round (a plain loop was one round trip per block)
      float v0[16], v5[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = min(b0 + u, nblk - 1);
        v0[u] = wk.GBP[(size_t) b * SPW + tid], v5[u] = wk.GBP[((size_t) nblk + b) * SPW + tid];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (b0 + u < nblk) gb0 += v0[u], gb5 += v5[u];
    }
    const float* w0 = n.W[0] + (size_t) tid * IN0 + PDIM;
    const float* w5 = n.W[SKIP + 1] + (size_t) tid * (IN0 + SPW) + PDIM;
    float pr[TOUT];
#pragma unroll
    for (int c = 0; c < TOUT; ++c) pr[c] = gb0 * w0[c] + gb5 * w5[c];
#pragma unroll
    for (int c = 0; c < TOUT; ++c) s_prod[c * SPW + tid] = pr[c];
  }
  __syncthreads();
  if (tid < TOUT * 8) {
    const int c = tid >> 3, part = tid & 7;
    float v = 0.f;
    for (int o = part; o < SPW; o += 8) v += s_prod[c * SPW + o];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (part == 0) s_gt[c] = v;
  }
  __syncthreads();
  if (act) {  // second linear: gW2 [30][256] = g_t (x) hid, gb2 = g_t;  g_hid = W2^T g_t * (hid > 0)
    const float hid = sv.thid[tid];
    float gh = 0.f;
    float w2[TOUT];
#pragma unroll
    for (int c = 0; c < TOUT; ++c) w2[c] = n.tw2[c * THID + tid];  // (one round of loads)
#pragma unroll
    for (int c = 0; c < TOUT; ++c) {
      g.tw2[c * THID + tid] = s_gt[c] * hid;
      gh += s_gt[c] * w2[c];
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

// ================================================================================================ backward, launch B
// job table: [0,112) hidden x hidden products of layers 1..7 (16 tiles of 64 x 64 each; layer 5 writes at column 93),
// [112,120) layer 0 (256 x 93: 4 x 2 tiles), [120,128) layer 5's input part (256 x 93), [128,132) heads (10 x 256: 4 tiles of
// 16 x 64).  Rows (the contraction) split over the 4 waves, partial tiles summed through LDS.
constexpr double ROWS_LAUNCH_SHARE = 0.56;  // of the optimizer's side range: beside launch A; the rest beside launch B
constexpr size_t WEIGHTS_LDS_BYTES = ((size_t) NWB * 64 * 65 + NWB * 64) * 4;
constexpr int JOBS_HH = 7 * 16, JOBS_X0 = 8, JOBS_HEAD = 4, N_JOBS = JOBS_HH + 2 * JOBS_X0 + JOBS_HEAD;

__global__ void __launch_bounds__(NTB) sp_net_backward_weights_kernel(int M, int n_jobs, NetPtrs n, GradPtrs g, SavedView sv,
    WorkView wk, SideAdam side) {
  if ((int) blockIdx.x >= n_jobs) {  // the CUs the 133 jobs leave idle: the second piece of the optimizer's side range
    side_adam_walk(side, (int) blockIdx.x - n_jobs, (int) gridDim.x - n_jobs);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float s_dynb[];  // [NWB][64 * 65] partial tiles | [NWB][64] column sums
  float (*s_part)[64 * 65] = reinterpret_cast<float (*)[64 * 65]>(s_dynb);
  float (*s_gb)[64]        = reinterpret_cast<float (*)[64]>(s_dynb + NWB * 64 * 65);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int Mp  = pad_rows(M);
  const int job = blockIdx.x;
  if (job == N_JOBS) {  // one more workgroup: the time network's backward (needs nothing from the other jobs)
    if (n.tw1) timenet_backward(M, n, g, sv, wk, &s_part[0][0]);  // (raw time encoding: the time is data, nothing to do)
    return;
  }
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
    G = g.W[layer], gld = layer_ld(layer, n.in0), gofs = layer_hofs(layer, n.in0);
    bias = (t % 4) == 0;
  } else if (job < JOBS_HH + 2 * JOBS_X0) {
    const int t = (job - JOBS_HH) % JOBS_X0;
    layer = job < JOBS_HH + JOBS_X0 ? 0 : SKIP + 1;
    o0 = 64 * (t / 2), k0 = 64 * (t % 2), kvalid = n.in0;
    X = sv.x0, xld = IN0P;
    A = wk.GZ + (size_t) layer * Mp * SPW;
    G = g.W[layer], gld = layer_ld(layer, n.in0), gofs = 0;
    bias = layer == 0 && (t % 2) == 0;
  } else {
    heads = true, layer = SPD, o0 = 0, k0 = 64 * (job - JOBS_HH - 2 * JOBS_X0), kvalid = SPW;
    X = sv.Y + (size_t) (SPD - 1) * Mp * SPW, xld = SPW;
    A = wk.GH;
    G = nullptr, gld = SPW, gofs = 0;
    bias = k0 == 0;
  }
  // ---- the wave's share of the rows: rows [rb, re), 4 per step
  const int per = ((Mp + NWB - 1) / NWB + 3) / 4 * 4;
  const int rb = min(wave * per, Mp), re = min(rb + per, Mp);
  f32x4 acc[4][4];  // [o tile a][k tile c]: outputs o0 + 4 i + a (heads: o = i), k0 + 4 j + c
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int kcol = k0 + 4 * j;
  // the loads of BATCH steps (4 rows each) are in flight before their MFMAs, the next batch's behind them (a load -> use chain
  // per step was 32 dependent L2 round trips: 50 us for this launch)
  constexpr int BATCH = 8;
  if (!heads) {
    const float* ap = A + (size_t) q * SPW + o0 + 4 * j;   // lane (i = j, kk = q)
    const float* xp = X + (size_t) q * xld + kcol;
    const bool xin  = kcol < xld;
    float4 av[2][BATCH], bv[2][BATCH];
    auto fetch = [&](int r, float4 (&a)[BATCH], float4 (&b)[BATCH]) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int rr  = r + 4 * u;
        const bool in = rr < re;
        const int rc  = in ? rr : rb;  // (clamped address, value masked: a select between POINTERS put the zero in scratch)
        a[u] = *reinterpret_cast<const float4*>(ap + (size_t) rc * SPW);
        b[u] = *reinterpret_cast<const float4*>(xin ? xp + (size_t) rc * xld : ap + (size_t) rc * SPW);
        if (!in) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(in && xin)) b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto consume = [&](const float4 (&aa)[BATCH], const float4 (&bb)[BATCH]) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const float4 a = aa[u], b = bb[u];
        colsum.x += a.x, colsum.y += a.y, colsum.z += a.z, colsum.w += a.w;
        acc[0][0] = mfma4(a.x, b.x, acc[0][0]), acc[0][1] = mfma4(a.x, b.y, acc[0][1]);
        acc[0][2] = mfma4(a.x, b.z, acc[0][2]), acc[0][3] = mfma4(a.x, b.w, acc[0][3]);
        acc[1][0] = mfma4(a.y, b.x, acc[1][0]), acc[1][1] = mfma4(a.y, b.y, acc[1][1]);
        acc[1][2] = mfma4(a.y, b.z, acc[1][2]), acc[1][3] = mfma4(a.y, b.w, acc[1][3]);
        acc[2][0] = mfma4(a.z, b.x, acc[2][0]), acc[2][1] = mfma4(a.z, b.y, acc[2][1]);
        acc[2][2] = mfma4(a.z, b.z, acc[2][2]), acc[2][3] = mfma4(a.z, b.w, acc[2][3]);
        acc[3][0] = mfma4(a.w, b.x, acc[3][0]), acc[3][1] = mfma4(a.w, b.y, acc[3][1]);
        acc[3][2] = mfma4(a.w, b.z, acc[3][2]), acc[3][3] = mfma4(a.w, b.w, acc[3][3]);
      }
    };
    fetch(rb, av[0], bv[0]);
    for (int r = rb; r < re; r += 8 * BATCH) {
      if (r + 4 * BATCH < re) fetch(r + 4 * BATCH, av[1], bv[1]);
      __builtin_amdgcn_sched_barrier(0);
      consume(av[0], bv[0]);
      if (r + 4 * BATCH < re) {
        if (r + 8 * BATCH < re) fetch(r + 8 * BATCH, av[0], bv[0]);
        __builtin_amdgcn_sched_barrier(0);
        consume(av[1], bv[1]);
      }
    }
  } else {
    const float* ap = A + (size_t) q * 16 + j;  // GH row r + q, output column j (>= 10: zero)
    const float* xp = X + (size_t) q * xld + kcol;
    for (int r = rb; r < re; r += 4 * BATCH) {
      float av[BATCH];
      float4 bv[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int rr  = r + 4 * u;
        const bool in = rr < re;
        const int rc  = in ? rr : rb;
        av[u] = ap[(size_t) rc * 16];
        bv[u] = *reinterpret_cast<const float4*>(xp + (size_t) rc * xld);
        if (!in) av[u] = 0.f, bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        colsum.x += av[u];
        acc[0][0] = mfma4(av[u], bv[u].x, acc[0][0]);
        acc[0][1] = mfma4(av[u], bv[u].y, acc[0][1]);
        acc[0][2] = mfma4(av[u], bv[u].z, acc[0][2]);
        acc[0][3] = mfma4(av[u], bv[u].w, acc[0][3]);
      }
    }
  }
  // ---- every wave parks its partial tile in LDS ([o local (64)][k local (64)], pitch 65); then ALL threads add the NWB partials,
  // 8 outputs each, and store rows of 64 consecutive columns.  D layout: row 4 q + r of tile a <-> o local 4 (4 q + r) + a
  // (heads: 4 q + r), column j of tile c <-> k local 4 j + c
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
  {
    float* sp = s_part[wave];
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
  const int no = heads ? 16 : 64;
  for (int e = tid; e < no * 64; e += NTB) {
    const int ol = e >> 6, kl = e & 63;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWB; ++w) v += s_part[w][ol * 65 + kl];
    if (k0 + kl >= kvalid) continue;
    if (heads) {
      if (ol < n.nout) {
        int hrow;
        const int hd = head_of(ol, hrow);
        g.head_w[hd][(size_t) hrow * SPW + k0 + kl] = v;
      }
    } else {
      G[(size_t) (o0 + ol) * gld + gofs + k0 + kl] = v;
    }
  }
  if (bias && tid < no) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWB; ++w) v += s_gb[w][tid];
    if (heads) {
      if (tid < n.nout) {
        int hrow;
        const int hd = head_of(tid, hrow);
        g.head_b[hd][hrow] = v;
      }
    } else {
      g.b[layer][o0 + tid] = v;
    }
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
  n.head_w[3] = d->local_w, n.head_b[3] = d->local_b;
  n.nout  = (d->local_w && d->local_b) ? NOUT_MAX : NOUT;
  n.lbs_c = (d->flags & SKGS_SP_NET_LBS_C) ? 1 : 0;
  n.tdim  = (d->flags & SKGS_SP_NET_RAW_TIME) ? 1 + 2 * ((d->flags >> 8) & 0xff) : 0;
  n.in0   = n.tdim ? PDIM + n.tdim : IN0;
  if (n.tdim) n.tw1 = n.tb1 = n.tw2 = n.tb2 = nullptr;
  return n;
}
bool net_complete(const skgs_sp_net* d) {
  bool ok = d->warp_w && d->warp_b && d->scaling_w && d->scaling_b && d->rotation_w && d->rotation_b;
  if (d->flags & SKGS_SP_NET_RAW_TIME) ok = ok && ((d->flags >> 8) & 0xff) <= 15;  // 63 + 1 + 2 * 15 = 94 <= 96 padded columns
  else ok = ok && d->time_w1 && d->time_b1 && d->time_w2 && d->time_b2;
  for (int l = 0; l < SPD; ++l) ok = ok && d->W[l] && d->b[l];
  return ok;
}
int allow_weights_lds() {  // the weight-gradient launch parks 8 partial 64 x 64 tiles: more than the default 64 KB of LDS
  static int rc = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(sp_net_backward_weights_kernel),
               hipFuncAttributeMaxDynamicSharedMemorySize, (int) WEIGHTS_LDS_BYTES) == hipSuccess ? 0 : 1;
  }();
  return rc;
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
    size_t saved_bytes, const skgs_sp_prepare* prepare, skgs_stream_t stream) {
  SKGS_REQUIRE(net && net->M >= 0, "sp_net_forward: NULL descriptor or M < 0");
  if (net->M == 0) return 0;
  SKGS_REQUIRE(net->points && net->time && net_complete(net), "sp_net_forward: NULL points / time / parameter");
  SKGS_REQUIRE(saved && saved_bytes >= skgs_sp_net_saved_bytes(net->M), "sp_net_forward: saved buffer too small");
  SKGS_REQUIRE(raw || bone_T, "sp_net_forward: no output requested");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_SP_NET_FWD, s);
  const SavedView sv = saved_view(saved, net->M);
  const NetPtrs n    = net_ptrs(net);
  SpPrepareJob prep{};
  int prep_groups = 0;
  if (prepare && prepare->pairs) {
    SKGS_REQUIRE(prepare->M >= 1 && prepare->K >= 1 && (prepare->F == 0 || prepare->F == 8) && prepare->sp_points &&
                     (prepare->F == 0 || prepare->sp_feature),
        "sp_net_forward: bad prepare job");
    SKGS_REQUIRE(prepare->pairs_bytes >= skgs_sp_pairs_bytes(prepare->P > 0 ? prepare->P : 1, prepare->M, prepare->K),
        "sp_net_forward: pair-list buffer too small (skgs_sp_pairs_bytes)");
    const SpPairsView pv = sp_pairs_view(prepare->pairs, prepare->P > 0 ? prepare->P : 1, prepare->M, prepare->K);
    const int n_clear = 64 + (prepare->M + 63) / 64 * 64;
    prep = SpPrepareJob{prepare->M, prepare->F, prepare->sp_points, prepare->sp_feature, prepare->sp_order, pv.header, n_clear, pv.table};
    prep_groups = (std::max(n_clear, prepare->M * 12) + 255) / 256;
  }
  hipLaunchKernelGGL(sp_net_transpose_kernel, dim3(TRANSPOSE_GROUPS + prep_groups), dim3(256), 0, s, n, sv.wt, prep);
  SKGS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sp_net_forward_kernel, dim3(pad_rows(net->M) / RB), dim3(NT), 0, s, net->M, n, raw, bone_T, d_rot, d_scale, sv);
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
  const int M = net->M, nblk = pad_rows(M) / RB;
  // The optimizer piece rides on BOTH launches, split like their durations (the row blocks' walk ~24 us on 128 CUs, the weight
  // gradients ~19 us on 133): alone beside the first it took 47 us and made that launch the step's long pole.
  SideAdam sd{}, sd2{};
  int n_side = 0, n_side2 = 0;
  const int n_jobs = N_JOBS + 1;
  if (side && side->n_tensors > 0) {
    SKGS_REQUIRE(side->tensors && side->step_count && side->chunk_begin >= 0 && side->chunk_end >= side->chunk_begin,
        "sp_net_backward: bad side range");
    sd.tensors = reinterpret_cast<const AdamTensor*>(side->tensors), sd.n = side->n_tensors;
    sd.c0 = side->chunk_begin, sd.c1 = side->chunk_end;
    sd.beta1 = side->beta1, sd.beta2 = side->beta2, sd.eps = (float) side->eps;
    sd.state = reinterpret_cast<const AdamState*>(side->step_count), sd.after_advance = side->after_advance ? 1 : 0;
    sd2 = sd;
    const int64_t cut = sd.c0 + (int64_t) ((sd.c1 - sd.c0) * ROWS_LAUNCH_SHARE);
    sd.c1 = cut, sd2.c0 = cut;
    auto side_groups = [](const SideAdam& a, int busy) {  // one workgroup (two 256-thread halves, two chunks each per iteration) per idle CU
      return a.c1 > a.c0 ? (int) std::max<long long>(1, std::min<long long>((a.c1 - a.c0 + 3) / 4, (long long) std::max(cu_count() - busy, 1))) : 0;
    };
    n_side = side_groups(sd, nblk), n_side2 = side_groups(sd2, n_jobs);
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
  g.head_w[3] = const_cast<float*>(grads->local_w), g.head_b[3] = const_cast<float*>(grads->local_b);
  g.points = const_cast<float*>(grads->points);
  SKGS_REQUIRE(n.nout == NOUT || (g.head_w[3] && g.head_b[3]), "sp_net_backward: the local-rotation head needs gradient pointers too");
  SKGS_REQUIRE(!n.lbs_c || net->points, "sp_net_backward: LBS_c needs the superpoint positions");
  ProfScope prof(K_SP_NET_BWD, s);
  SKGS_REQUIRE(allow_weights_lds() == 0, "sp_net_backward: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(sp_net_backward_rows_kernel, dim3(nblk + n_side), dim3(NT), 0, s, M, nblk, n, g_bone_T, g_d_rot, g_d_scale, g_raw,
      n.lbs_c ? g.points : nullptr, sv, wk, sd);
  SKGS_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sp_net_backward_weights_kernel, dim3(n_jobs + n_side2), dim3(NTB), WEIGHTS_LDS_BYTES, s, M, n_jobs, n, g, sv, wk, sd2);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
