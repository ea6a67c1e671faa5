// mlp_fused.hip -- the bone-transform producer (scope row (f)-3) as ONE persistent launch per direction.
//
// Reference: SimpleDeformationNetwork (networks/sk_gs.py:134-164) = FreqEncoder(joints, degree 10) | FreqEncoder(t,
// degree 6) -> MLP_with_skips (my_ext/blocks/mlp.py:43-85: 8 x 256, ReLU, the encoded input concatenated again after
// layer 4) -> heads 4 | 4 | 3, evaluated on ONE ROW PER BONE (B = M ~ 20..32) in every training step
// (networks/sk_gs.py:1073-1079).  torch runs it as ~65 launches; mlp.hip as one launch per layer and direction (24
// launches, 8.6 / 13.5 us each: each launch pays a boundary, a global -> LDS staging round trip and a serial dot
// product).  The arithmetic is nothing (20 x 256 x 256 FMAs per layer); the 2.1 MB of weights and the NINE DEPENDENT
// LAYERS are the cost.  So:
//
//   * G = H / NC = 32 workgroups of 512 threads (NC = 8 output features each), one per CU, all resident.  Workgroup g owns output features [g NC, (g+1) NC) of EVERY layer: its rows of every
//     weight matrix are fetched ONCE, up front, with all loads in flight, into LDS -- the 2.1 MB stream is spread over the
//     workgroups instead of 9 dependent staging round trips.
//   * between layers the [B, H] activations are exchanged IN-LAUNCH.  Every workgroup owns one contiguous slab
//     [Bp][NC] of the exchange image of a layer and publishes it with ONE wave-wide 16-byte write-through (sc1) store
//     instruction; every workgroup reads all slabs with 16-byte agent-scope (sc1) loads.  Each 4-byte word validates
//     itself: the image is pre-filled with a sentinel (0xFFFFFFFF, a NaN pattern no arithmetic produces) and a reader
//     re-reads until none of its words is the sentinel -- no flag, no fence, no release (the data-tagged-granule
//     hand-off of MI355X_MICROARCH.md "visibility", with the tag folded into the value: half the bytes on the wire).
//     Two images per layer alternate with the launch parity; a launch re-fills its slabs of the OTHER image (read by
//     nobody in this launch) with the sentinel, the kernel boundary publishes that.  The launch counters live in the
//     workspace header, so a hipGraph replay needs no memset node and no per-launch argument.
//   * every spin is bounded (50 ms of s_memrealtime AND 100 000 executed polls): a launch that cannot complete sets a sticky failure word instead of
//     hanging the device.
//   * backward: workgroup g owns INPUT features [g NC, ..) of every layer (= the output features it owned one layer
//     earlier): gA_{l-1}[:, slab] = gZ_l W_l[:, slab] is the only thing on the dependent chain; the weight gradient of
//     its own rows, gW_l[slab, :] = gZ_l[:, slab]^T [a_{l-1} | x0], needs nothing from other workgroups but the saved
//     activations and runs while the other workgroups' slabs are in flight.
//
// A layer's product [B rows] x [NC = 8 columns] x [K = 256 (+ encoded input)] runs on the matrix cores as outer products:
// v_mfma_f32_4x4x1_16B_f32 holds 16 independent 4 x 4 blocks per wave = 8 row groups x 2 column groups, one k per
// instruction; the 8 waves of the workgroup split K (32 consecutive k each: four ds_read_b128 per operand and row), their
// partial blocks are summed through LDS in a fixed order by the threads that publish the slab.  (The first version did
// this on the VALU -- a lane per (row, 4 columns, 16 k), DPP row sums: 24 ds_read_b128 + 160 VALU instructions per lane and
// layer, 3000 clocks at two waves per SIMD; the MFMA form issues 32 matrix + 16 LDS instructions.)  Row pitches in LDS are
// padded by 4 floats so that the 16 rows a ds_read_b128 touches start in different banks.
#include <algorithm>
#include <cstdlib>

#include "adam_update.h"
#include "bone_chain.inl"
#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int MAXL = SKGS_MLP_MAX_LAYERS;
constexpr unsigned SENTINEL = 0xffffffffu;
// Shapes fixed at compile time: hidden width H = 256, encoded input padded to INP = 128, at most KL layers.
constexpr int H = 256, INP = 128, KL = 10;
constexpr int NT = 512, NW = NT / 64, NC = 8;  // threads, waves and output features per workgroup
constexpr int G_NET = H / NC;                   // the network's workgroups
constexpr int HP = H + 4;                       // LDS row pitch of a [rows][H] image (see the thread map above)

using gu32 = __attribute__((address_space(1))) unsigned int;
typedef float f4 __attribute__((ext_vector_type(4)));  // a VGPR quad as an asm operand

struct FusedLayer {
  const float* W;     // [out, in_h + in_x]
  const float* bias;  // [out]
  float* gW;
  float* gb;
  int in_h, in_x, out, relu;
};
struct FusedArgs {
  int B, p_dim, p_deg, t_dim, t_deg, IN, n_layers, lds_floats;
  const float* points;
  const float* t;
  float* x0;          // forward: [B, IN] (written) or NULL; backward: the forward's copy (read) or NULL (re-encoded)
  float* acts;        // forward: written; backward: read.  [n_layers - 1][B][H]
  float* out;         // forward: [B, out_last]
  const float* g_out; // backward: [B, out_last]
  float* g_x0;        // backward: [B, IN] or NULL
  unsigned* hdr;      // workspace header: [0] forward launches, [1] failed launches (sticky), [2] stamps wanted,
                      // [3] backward launches, [4] / [5] forward / backward launches whose exchange ran on plain stores
                      // (xcd_mode 1: the census found the network on one XCD), [6] / [7] bit x: workgroup 0 of a forward /
                      // backward launch has run on XCD x, [8..14] the frame's row of global_T (skeleton
                      // forward -> backward), [16..63] stamps
  float* exch;        // this direction's exchange images [2][n_layers - 1][G][Bp][NC]
  int n_heads, head_dim[4];  // the last layer's columns split over separate [B, head_dim[j]] tensors (n_heads = 0: one tensor)
  float* head_out[4];        // forward
  const float* head_gout[4]; // backward
  // per-layer data as parallel arrays, not an array of structs: the prologue needs every W / bias pointer at once, and ten
  // 48-byte descriptors exceed the SGPR file -- hipcc then serialises one scalar load + wait per layer, each a round trip to
  // the (host-visible) kernarg segment: 2.9 us before the first weight load was issued
  const float* W[KL];
  const float* bias[KL];
  float* gW[KL];
  float* gb[KL];
  unsigned xmask, relu_mask;  // bit l: layer l reads the encoded input / applies ReLU
  int out_last;
  // side job of the backward launch: its workgroups beyond the network's G apply the Adam update of the chunks
  // [adam_c0, adam_c1) of an optimizer table (adam.hip) -- see adam_side_job
  const AdamTensor* adam_tensors;
  int adam_n;
  long long adam_c0, adam_c1;
  double adam_beta1, adam_beta2;
  float adam_eps;
  const AdamState* adam_step;
  int adam_after_advance;  // the side piece follows the launch that advanced the counter (see adam_coefficients)
  // the kinematic chain riding on the launches (skgs_skeleton_forward / _backward): forward -- workgroup 0, which owns
  // the raw joint rotations (head 0 = the last layer's columns 0..3), runs it after the heads; backward -- every workgroup
  // runs its backward in the prologue (the gradient of head 0 is the one input of the network's backward no other kernel
  // has produced yet)
  int has_chain;
  chain::ChainArgs chain;
  // skeleton forward in training: the frame's row of the test-time cache, sk_cache[frame_index] = [normalised joint
  // rotation | d_rot | d_scale] (networks/sk_gs.py:1077-1079), written by the workgroups that own the heads' columns
  float* sk_cache;  // [frames][B][out_last] or NULL
  // Where the network's workgroups run.
  // 0: blocks 0 .. G-1 (the dispatcher deals them four to each XCD); slabs leave as write-through (sc1) stores.
  // 1: blocks 0, 8, 16 .. 8 (G - 1) -- blocks of one residue mod 8 are observed to share an XCD (MI355X_MICROARCH.md, dispatch), so the
  //    whole exchange can stay inside ONE L2: slabs leave as PLAIN stores (the line stays in that L2; an sc1 store drops it), readers
  //    poll with sc1 loads (L1 bypassed) as before.  Placement is no contract: every workgroup publishes the XCC_ID it runs on next to
  //    its first slab (always write-through), every workgroup reads all of them with its first gather, and only a launch whose
  //    network sits on ONE XCD switches to plain stores -- the same 32 words are seen by all, so all decide alike.  A launch placed
  //    otherwise runs on write-through stores as in mode 0.
  // 2: the placement of 1, write-through stores throughout (A/B of the placement alone).
  // 3: mode 1 with a FALSIFIED census (tests: the fall-back path of a launch that is not on one XCD).
  int xcd_mode;
  int side_delay;  // the side job's workgroups start this many x ~0.85 us late (the network's weight loads go first)
  unsigned* census;  // this direction's [2 parities][G][4 words]: (XCC_ID, 0, 0, 0) per network workgroup, sentinel-filled like the images
};

__device__ __forceinline__ FusedLayer get_layer(const FusedArgs& a, int l) {
  return FusedLayer{a.W[l], a.bias[l], a.gW[l], a.gb[l], l ? H : 0, ((a.xmask >> l) & 1u) ? a.IN : 0,
      l == a.n_layers - 1 ? a.out_last : H, (int) ((a.relu_mask >> l) & 1u)};
}

__device__ __host__ __forceinline__ int pad32(int x) { return (x + 31) & ~31; }

// element (row, col) of the last layer's output / incoming gradient: one [B, out] tensor or one tensor per head
__device__ __forceinline__ float* head_elem(const FusedArgs& a, float* const* heads, float* single, int row, int col, int out) {
  if (a.n_heads == 0) return single + (size_t) row * out + col;
  int j = 0, off = 0;
  while (j < a.n_heads - 1 && col >= off + a.head_dim[j]) off += a.head_dim[j++];
  return heads[j] + (size_t) row * a.head_dim[j] + (col - off);
}

__device__ __forceinline__ float row_sum_to_lane15(float v) {
  v += dpp_mov<0x111, 0xf, 0xf, true>(v);  // row_shr:1
  v += dpp_mov<0x112, 0xf, 0xf, true>(v);  // row_shr:2
  v += dpp_mov<0x114, 0xf, 0xf, true>(v);  // row_shr:4
  v += dpp_mov<0x118, 0xf, 0xf, true>(v);  // row_shr:8 -> lane 15 of each 16-lane row = row sum
  return v;
}

// (the frequency encoding of freqencoder.cu:7-33 is inlined in the prologues: raw inputs are loaded first, the sines are
// taken while the weight loads are in flight; same expression as mlp.hip::freq_encode_forward_kernel)

// diagnostics (header word 2 != 0): workgroup 0 records {100 MHz real-time counter, shader clock counter} at successive
// points of the launch into header words 16.. (two words per stamp, 24 stamps)
__device__ __forceinline__ void stamp(const FusedArgs& a, const unsigned* s_misc, int& si, unsigned long long t_entry = 0) {
  if (s_misc[2] && blockIdx.x == 0 && threadIdx.x == 0 && si < 20) {
    if (t_entry) a.hdr[16 + 46] = (unsigned) t_entry;
    a.hdr[16 + 2 * si]     = (unsigned) __builtin_amdgcn_s_memrealtime();
    a.hdr[16 + 2 * si + 1] = (unsigned) __builtin_amdgcn_s_memtime();
    ++si;
  }
}

// head_elem with every pointer and width read at a STATIC index (scalar kernel-argument loads that go out with the launch's first
// batch) and selected by compares: a dynamic index into head_gout[] is a scalar load of its own from the host-visible argument
// segment, a ~3 us round trip in front of the load it feeds
__device__ __forceinline__ const float* head_elem_sel(const FusedArgs& a, const float* const* heads, const float* single, int row, int col,
    int out) {
  if (a.n_heads == 0) return single + (size_t) row * out + col;
  const int d0 = a.head_dim[0], d1 = a.head_dim[1], d2 = a.head_dim[2], d3 = a.head_dim[3];
  const float *h0 = heads[0], *h1 = heads[1], *h2 = heads[2], *h3 = heads[3];
  const int n = a.n_heads;
  const bool in0 = n == 1 || col < d0, in1 = n == 2 || col < d0 + d1, in2 = n == 3 || col < d0 + d1 + d2;
  const float* base = in0 ? h0 : in1 ? h1 : in2 ? h2 : h3;
  const int dim     = in0 ? d0 : in1 ? d1 : in2 ? d2 : d3;
  const int off     = in0 ? 0 : in1 ? d0 : in2 ? d0 + d1 : d0 + d1 + d2;
  return base + (size_t) row * dim + (col - off);
}

// diagnostics switch as every thread sees it (header word 2; a uniform load)
__device__ __forceinline__ bool stamps_on_all(const FusedArgs& a) { return a.hdr[2] != 0; }

// a produced value must never look like the "not written yet" pattern
__device__ __forceinline__ float not_sentinel(float v) { return f2u(v) == SENTINEL ? u2f(0x7fc00000u) : v; }

// 16-byte write-through store (global_store_dwordx4 ... sc1); the trailing s_nop keeps hipcc's next instruction from
// overwriting the data registers before the store has read them (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void store16_sc1(float* p, float4 v) {
  const f4 q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}

// plain 16-byte store: only for readers behind the SAME L2 (xcd_mode 1)
__device__ __forceinline__ void store16_plain(float* p, float4 v) {
  const f4 q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void store16_exch(float* p, float4 v, bool plain) {
  if (plain) store16_plain(p, v);
  else store16_sc1(p, v);
}

// Read every workgroup's slab of one exchange image into the [Bp][H] LDS image (row pitch H), re-reading until no word
// is the sentinel.  Unit u = 16 bytes: slab g = u / (B NC/4), then row-major inside the slab (rows < B only).  All of a
// thread's loads are in flight together (one asm statement with its own wait: hipcc does not track asm loads).
// Returns false on time-out.
// `census`: this launch's [G][4] census words; lane i (mod 32) also fetches workgroup i's entry in the same batch of loads and holds
// it back in `census_word` (valid, like the slabs, once the function returns true).
template <int U>
__device__ __forceinline__ bool gather_slabs(const float* img, float* s_dst, int B, int Bp, int pitch, int n_units, const unsigned* census,
    unsigned& census_word) {
  constexpr int Q = NC / 4;  // 16-byte units per slab row
  const unsigned* cptr = census + 4 * (threadIdx.x & (G_NET - 1));
  f4 cv;
  const float* ptr[U];
  int dst[U];
  bool live[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    int u   = j * NT + (int) threadIdx.x;
    live[j] = u < n_units;
    if (!live[j]) u = 0;
    const int g = u / (B * Q), rem = u - g * (B * Q), row = rem / Q, part = rem - row * Q;
    ptr[j] = img + ((size_t) g * Bp + row) * NC + 4 * part;
    dst[j] = row * pitch + g * NC + 4 * part;
  }
  f4 v[U];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool good = true;
  unsigned polls = 0;
  for (;;) {
    if constexpr (U == 4) {
      asm volatile(
          "global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\t"
          "global_load_dwordx4 %2, %7, off sc1\n\tglobal_load_dwordx4 %3, %8, off sc1\n\t"
          "global_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(cv)
          : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]), "v"(cptr)
          : "memory");
    } else {
      static_assert(U == 4 || U == 8, "gather_slabs: 4 or 8 units per thread");
      asm volatile(
          "global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %10, off sc1\n\t"
          "global_load_dwordx4 %2, %11, off sc1\n\tglobal_load_dwordx4 %3, %12, off sc1\n\t"
          "global_load_dwordx4 %4, %13, off sc1\n\tglobal_load_dwordx4 %5, %14, off sc1\n\t"
          "global_load_dwordx4 %6, %15, off sc1\n\tglobal_load_dwordx4 %7, %16, off sc1\n\t"
          "global_load_dwordx4 %8, %17, off sc1\n\ts_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4 % U]), "=&v"(v[5 % U]), "=&v"(v[6 % U]), "=&v"(v[7 % U]), "=&v"(cv)
          : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]), "v"(ptr[4 % U]), "v"(ptr[5 % U]), "v"(ptr[6 % U]), "v"(ptr[7 % U]), "v"(cptr)
          : "memory");
    }
    bool ok = f2u(cv.x) != SENTINEL;
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (live[j])
        ok &= f2u(v[j].x) != SENTINEL && f2u(v[j].y) != SENTINEL && f2u(v[j].z) != SENTINEL && f2u(v[j].w) != SENTINEL;
    if (ok) break;
    // give up after 50 ms on the wall clock AND 100 000 polls actually EXECUTED (>= 0.1 s of spinning: a poll is a memory round trip).
    // The wall clock alone is not a measure of waiting: s_memrealtime keeps running while the queue is switched out (CWSR) -- eight
    // processes time-sharing one GPU (tests/test_gpu_bench_contract.py: 8 ranks on the one device) exceed 50 ms between two time
    // slices of a rank, and its launch "gave up" although every workgroup it waited for was merely not running (2 of 6 runs).
    if (++polls > 100000u && __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {
      good = false;
      break;
    }
  }
#pragma unroll
  for (int j = 0; j < U; ++j)
    if (live[j]) *reinterpret_cast<float4*>(s_dst + dst[j]) = make_float4(v[j].x, v[j].y, v[j].z, v[j].w);
  census_word = f2u(cv.x);
  return good;
}
// every lane holds one workgroup's census word (lane i mod 32: workgroup i): does the whole network run on one XCD?  Wave-uniform, and
// the same answer in every wave of every workgroup (all of them read the same G words).
__device__ __forceinline__ bool census_one_xcd(unsigned census_word) {
  const unsigned first = (unsigned) __builtin_amdgcn_readfirstlane((int) census_word);
  return __builtin_amdgcn_ballot_w64(census_word == first) == ~0ull;
}
// the XCD this wave runs on: HW_REG_XCC_ID (id 20), bits [3:0]
__device__ __forceinline__ unsigned xcc_id() { return (unsigned) __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)); }
// this workgroup's entry of the launch's census (thread 0; write-through: readers on any XCD)
__device__ __forceinline__ void census_publish(const FusedArgs& a, unsigned parity, int g) {
  if (threadIdx.x == 0) {
    unsigned id = xcc_id();
    if (a.xcd_mode == 3) id = (unsigned) (g & 1);  // (tests: a network that is NOT on one XCD)
    store16_sc1(reinterpret_cast<float*>(a.census + ((size_t) parity * G_NET + g) * 4), make_float4(u2f(id), 0.f, 0.f, 0.f));
  }
}

// Lane map of the 4x4x1 MFMA (checked on gfx950): lane = 4 b + j holds A[4 b + i = j] and B[4 b + j] of block b; VGPR i of the
// result holds D_b[i][j].  Block b = 2 rgl + cg: row group rgl (rows 4 rg .. 4 rg + 3, rg = 8 round + rgl), column group cg.
struct LaneMap {
  int j, cg, rgl, colw;  // colw = 4 cg + j: the weight row this lane feeds as B operand
};
__device__ __forceinline__ LaneMap lane_map() {
  const int l = threadIdx.x & 63;
  return LaneMap{l & 3, (l >> 2) & 1, l >> 3, 4 * ((l >> 2) & 1) + (l & 3)};
}

// acc[rd] += sum over k in [k0, k0 + 4 nk4) of src[row][k] w[col][k] for this lane's blocks (two accumulators per round: the
// chain of dependent MFMAs is half as long).  rowc[rd]: the LDS row this lane reads as A operand in round rd.
template <int ROUNDS>
__device__ __forceinline__ void mfma_step(f4 (&acc)[ROUNDS][2], const float* __restrict__ xs, int pitch, const int (&rowc)[ROUNDS],
    const float* __restrict__ ws) {
  const float4 wv = *reinterpret_cast<const float4*>(ws);
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const float4 xv = *reinterpret_cast<const float4*>(xs + rowc[rd] * pitch);
    acc[rd][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.x, wv.x, acc[rd][0], 0, 0, 0);
    acc[rd][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.y, wv.y, acc[rd][1], 0, 0, 0);
    acc[rd][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.z, wv.z, acc[rd][0], 0, 0, 0);
    acc[rd][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.w, wv.w, acc[rd][1], 0, 0, 0);
  }
}
// NK4 > 0: that many float4 steps, unrolled (all LDS reads in flight together); NK4 = 0: nk4 steps, rolled
template <int ROUNDS, int NK4 = 0>
__device__ __forceinline__ void mfma_dot(f4 (&acc)[ROUNDS][2], const float* __restrict__ s_src, int pitch, const int (&rowc)[ROUNDS],
    const float* __restrict__ s_w, int wpitch, int colw, int k0, int nk4 = NK4) {
  const float* ws = s_w + colw * wpitch + k0;
  const float* xs = s_src + k0;
  if constexpr (NK4 > 0) {
#pragma unroll
    for (int q = 0; q < NK4; ++q) mfma_step<ROUNDS>(acc, xs + 4 * q, pitch, rowc, ws + 4 * q);
  } else {
#pragma unroll 1
    for (int q = 0; q < nk4; ++q) mfma_step<ROUNDS>(acc, xs + 4 * q, pitch, rowc, ws + 4 * q);
  }
}
template <int ROUNDS>
__device__ __forceinline__ void zero_acc(f4 (&acc)[ROUNDS][2]) {
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) acc[rd][0] = acc[rd][1] = f4{0.f, 0.f, 0.f, 0.f};
}
// one column of a saved activation matrix [B][H]: rows in groups of four under one wave-uniform test; rows >= B re-read row
// B - 1 (their gZ rows are zeros)
template <int Bp>
__device__ __forceinline__ void load_act_column(float (&av)[Bp], const float* __restrict__ ap, int B) {
#pragma unroll
  for (int b4 = 0; b4 < Bp; b4 += 4) {
    if (b4 < B) {
#pragma unroll
      for (int u = 0; u < 4; ++u) av[b4 + u] = ap[min(b4 + u, B - 1) * H];
    }
  }
}
// this wave's partial blocks -> s_part [NW][Bp][NC]
template <int ROUNDS>
__device__ __forceinline__ void write_partials(float* s_part, const f4 (&acc)[ROUNDS][2], const LaneMap& m, int wave, int Bp,
    int n_rg) {
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int rg = 8 * rd + m.rgl;
    if (rg < n_rg) {
      const f4 v = acc[rd][0] + acc[rd][1];
      float* d   = s_part + ((size_t) (wave * Bp + 4 * rg)) * NC + m.colw;
      d[0] = v[0], d[NC] = v[1], d[2 * NC] = v[2], d[3 * NC] = v[3];
    }
  }
}
// columns 4 part .. 4 part + 3 of row `row`: the waves' partials added in wave order
__device__ __forceinline__ float4 sum_partials(const float* s_part, int Bp, int row, int part) {
  float4 y = *reinterpret_cast<const float4*>(s_part + (size_t) row * NC + 4 * part);
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    const float4 p = *reinterpret_cast<const float4*>(s_part + ((size_t) (w * Bp + row)) * NC + 4 * part);
    y.x += p.x, y.y += p.y, y.z += p.z, y.w += p.w;
  }
  return y;
}

// fill this workgroup's slabs of the image the launch does NOT use with the sentinel
__device__ __forceinline__ void repoison(float* img_other, int nX, int G, int Bp, int g, unsigned* census_other) {
  const int slab4 = Bp * NC / 4;
  const float4 s  = make_float4(u2f(SENTINEL), u2f(SENTINEL), u2f(SENTINEL), u2f(SENTINEL));
  if (threadIdx.x == 0) reinterpret_cast<float4*>(census_other)[g] = s;
  for (int i = threadIdx.x; i < nX * slab4; i += NT) {
    const int l = i / slab4, q = i - l * slab4;
    reinterpret_cast<float4*>(img_other + ((size_t) l * G + g) * Bp * NC)[q] = s;
  }
}

// The backward launch needs 32 CUs for ~30 us and leaves 224 idle; the optimizer update of the per-Gaussian parameters (a
// pure stream: 28 B per element, 40 us at 100k Gaussians on the whole chip) does not depend on it.  Workgroups G.. of the
// SAME launch therefore walk the chunks of that update (two 256-thread halves per workgroup, two chunks each per
// iteration): a branch of a captured graph or a second stream would cost more in fork / join edges than it hides (DESIGN
// section 7), workgroups of one launch cost nothing.  The network's workgroups have the lowest ids and are dispatched first,
// so all of them are resident before the first side workgroup is placed; the LDS request of the launch keeps it at one
// workgroup per CU.
// the side job gets one workgroup per CU the network does not occupy: the device's CU count, asked once per device (256 on
// a whole MI355X; a partitioned / shared GPU reports what this process can use)
inline int num_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;  // (a racing thread writes the same value)
  }
  return cached[dev];
}
// which workgroup of the launch is what (see FusedArgs::xcd_mode)
struct Role {
  bool net;
  int g, wg, n_side;  // network index, or index among the n_side others
};
__device__ __forceinline__ Role role_of(const FusedArgs& a) {
  const int b = (int) blockIdx.x, n_side = (int) gridDim.x - G_NET;
  if (!a.xcd_mode) return Role{b < G_NET, b, b - G_NET, n_side};
  const bool low = b < 8 * G_NET;
  const bool net = low && (b & 7) == 0;
  const int before = low ? (b >> 3) + 1 : G_NET;  // network workgroups with a smaller block id (b itself not a network one)
  return Role{net, b >> 3, b - before, n_side};
}
__device__ __forceinline__ void adam_side_job(const FusedArgs& a, int wg, int n_side) {
  if (!a.adam_tensors || a.adam_c1 <= a.adam_c0) return;  // (xcd_mode without a side range: the blocks between the network's just leave)
  for (int i = 0; i < a.side_delay; ++i) __builtin_amdgcn_s_sleep(32);  // (experiment: 32 x 64 clocks ~ 0.85 us per unit)
  const int half = threadIdx.x >> 8, t256 = threadIdx.x & 255, lane = threadIdx.x & 63;
  const AdamCoef k = adam_coefficients(a.adam_beta1, a.adam_beta2, a.adam_eps, a.adam_step, a.adam_after_advance != 0);
  const AdamTensorLanes desc = adam_load_descriptors(a.adam_tensors, a.adam_n, lane);
  const int64_t first0 = lane < a.adam_n ? a.adam_tensors[lane].chunk0 : INT64_MAX;
  // Every workgroup gets an equal, contiguous share of the chunks (+-1): with a grid-stride loop of 4 chunks per
  // workgroup and iteration, 1890 chunks over 224 workgroups were 2 iterations for some and 3 for others -- the launch
  // ended with the stragglers (57 us; the network alone: 45).  Inside its share a 256-thread half takes two chunks at a time
  // (all 32 loads of a thread in flight: with one workgroup per CU -- the launch's LDS request -- the bytes in flight per
  // CU are what bounds the stream).
  const int64_t n_chunks = a.adam_c1 - a.adam_c0;
  const int64_t begin = a.adam_c0 + n_chunks * wg / n_side, end = a.adam_c0 + n_chunks * (wg + 1) / n_side;
#if SKGS_SIDE_CHUNKS == 1   // one chunk per half and iteration: 0.3407 -> 0.3395 ms per step against two (8 alternating runs each) -- fewer
                            // bytes in flight beside the network's hand-offs, loads and write-through stores interleaved per 4 KB
  for (int64_t chunk = begin + half; chunk < end; chunk += 2) {
    const int ti0 = adam_owner(a.adam_tensors, a.adam_n, first0, lane, chunk);
    const AdamTensor T0 = ti0 < 64 ? adam_descriptor_of(desc, ti0) : a.adam_tensors[ti0];
    adam_update_chunk(T0, (chunk - T0.chunk0) * ADAM_CHUNK, t256, k);
  }
  return;
#endif
  for (int64_t chunk = begin + 2 * half; chunk < end; chunk += 4) {
    const int ti0 = adam_owner(a.adam_tensors, a.adam_n, first0, lane, chunk);
    const AdamTensor T0 = ti0 < 64 ? adam_descriptor_of(desc, ti0) : a.adam_tensors[ti0];
    if (chunk + 1 < end) {
      const int ti1 = adam_owner(a.adam_tensors, a.adam_n, first0, lane, chunk + 1);
      const AdamTensor T1 = ti1 < 64 ? adam_descriptor_of(desc, ti1) : a.adam_tensors[ti1];
      adam_update_chunk2(T0, (chunk - T0.chunk0) * ADAM_CHUNK, T1, (chunk + 1 - T1.chunk0) * ADAM_CHUNK, t256, k);
    } else {
      adam_update_chunk(T0, (chunk - T0.chunk0) * ADAM_CHUNK, t256, k);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- forward
// LDS (floats): s_x0 [Bp][XP] | s_act [Bp][HP] | s_part [NW][Bp][NC] | slabs: layer l -> [NC][Kp_l + 4], Kp_l = (l ? H : 0) +
//               (in_x ? XW : 0) | bias [KL][NC] | misc (launch count, fail, stamps).  XW = the encoded width rounded up to
//               32 (every wave takes XW / 8 of its k), XP = XW + 4.
template <int PASSES>
__global__ void __launch_bounds__(NT) fused_mlp_forward_kernel(const FusedArgs a) {
  constexpr int Bp = 16 * PASSES, ROUNDS = (4 * PASSES + 7) / 8;
  constexpr int U = (Bp * 64 + NT - 1) / NT <= 4 ? 4 : 8;  // 16-byte units per thread of one gather
  constexpr int EQ = (Bp * INP + NT - 1) / NT;             // encoded-input entries per thread (padding included)
  static_assert((Bp * 64 + NT - 1) / NT <= 8, "gather_slabs covers at most 8 units per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Role role = role_of(a);
  if (!role.net) {  // (the forward launch can host an optimizer piece on its idle CUs too: see adam_side_job)
    adam_side_job(a, role.wg, role.n_side);
    return;
  }
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, wave = tid >> 6;
  const int G = G_NET, g = role.g, col0 = g * NC;
  const int B = a.B, nL = a.n_layers, nX = nL - 1, IN = a.IN;
  const int XW = pad32(IN), XP = XW + 4;
  float* s_x0   = smem;
  float* s_act  = s_x0 + Bp * XP;
  float* s_part = s_act + Bp * HP;
  float* s_raw  = s_part + NW * Bp * NC;  // [Bp][4]: the raw joint rotations (head 0), kept for the kinematic chain
  float* s_w    = s_raw + Bp * 4;
  unsigned* s_misc = reinterpret_cast<unsigned*>(smem + a.lds_floats - 4);
  float* s_bias    = smem + a.lds_floats - 4 - KL * NC;
  const LaneMap lm = lane_map();
  int rowc[ROUNDS];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) rowc[rd] = (8 * rd + lm.rgl < 4 * PASSES) ? 4 * (8 * rd + lm.rgl) + lm.j : lm.j;

  // ---- prologue: ONE memory round trip.  Issue order: launch counter, raw encoder inputs, biases, every weight row this
  // workgroup will ever need (compile-time layer index: the descriptor reads stay scalar kernarg loads); then the sines,
  // then the LDS stores.
  unsigned cnt = 0, stamps_on = 0;
  if (tid == 0) {
    const gu32* h = reinterpret_cast<const gu32*>((unsigned long long) a.hdr);
    cnt       = __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamps_on = __hip_atomic_load(h + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  chain::Prefetch cpf;  // the kinematic chain's inputs (workgroup 0 runs it after the heads): loaded with everything else
  int frame_pf = 0;     // the frame's row of the test-time cache (the workgroups that own the heads' columns)
  if (a.sk_cache && a.chain.frame_index && col0 < a.out_last) frame_pf = a.chain.frame_index[0];
  cpf.valid = false;
  if (a.has_chain && g == 0) cpf = chain::prefetch(a.chain, false, false);
  // encoded input entry (b, c): c = tid & 127 is the same for all of a thread's entries (INP = 128), so everything that
  // depends on the column -- which raw coordinate, which frequency, sine or cosine -- is computed once, without divisions
  // in the per-entry loop; b = tid / 128 + q NT / 128
  const int pe = a.p_dim * (1 + 2 * a.p_deg);
  const int ec = tid & (INP - 1), eb0 = tid >> 7;
  const bool e_live = ec < IN;
  const bool e_pt   = ec < pe;
  const int e_cc = e_pt ? ec : ec - pe, e_D = e_pt ? a.p_dim : a.t_dim;
  const int e_col = e_cc / e_D - 1, e_d = e_cc - (e_col + 1) * e_D;  // column group (-1: the raw value), coordinate
  const float e_phase = (float) (e_col & 1) * (3.141592653589793f / 2);
  float xin[EQ];
#pragma unroll
  for (int q = 0; q < EQ; ++q) {
    const int b = eb0 + q * (NT / INP);
    xin[q] = 0.f;
    if (e_live && b < B) xin[q] = e_pt ? a.points[(size_t) b * a.p_dim + e_d] : a.t[e_d];
  }
  float bv[KL];
  float4 vh[KL], vx[KL];
  const int ch = tid >> 6, kh = 4 * (tid & 63);  // hidden part: NC rows x 64 units = NT units
  const int cx = tid >> 5, kx = 4 * (tid & 31);  // x0 part: NC rows x 32 units = NT / 2 units
  {
    // every layer's pointers are fetched from the kernarg segment up front, in a few wide scalar loads and ONE wait (left
    // inside the per-layer branches they were ten dependent round trips to host-visible memory: 2.8 us)
    const float* wp[KL];
    const float* bp[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) wp[l] = a.W[l], bp[l] = a.bias[l];
#pragma unroll
    for (int l = 0; l < KL; ++l) asm volatile("" : "+s"(wp[l]), "+s"(bp[l]));
#pragma unroll
    for (int l = 0; l < KL; ++l) {
      bv[l] = 0.f;
      vh[l] = vx[l] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l < nL) {
        FusedLayer L = get_layer(a, l);
        L.W = wp[l], L.bias = bp[l];
        const int K = L.in_h + L.in_x;
        if (tid < NC && L.bias && col0 + tid < L.out) bv[l] = L.bias[col0 + tid];
        if (l > 0 && col0 + ch < L.out) vh[l] = *reinterpret_cast<const float4*>(L.W + (size_t) (col0 + ch) * K + kh);
        if (L.in_x && tid < NT / 2 && col0 + cx < L.out && kx < L.in_x)
          vx[l] = *reinterpret_cast<const float4*>(L.W + (size_t) (col0 + cx) * K + L.in_h + kx);
      }
    }
  }
  // the encoded input (the loads above are still in flight behind these); padding entries are written as zeros
#pragma unroll
  for (int q = 0; q < EQ; ++q) {
    const int b = eb0 + q * (NT / INP);
    if (b < Bp && ec < XW) {
      float v = xin[q];
      if (e_live && b < B && e_col >= 0) v = sinf(scalbnf(v, e_col >> 1) + e_phase);
      s_x0[b * XP + ec] = v;
    }
  }
  for (int i = tid; i < Bp * HP / 4; i += NT) reinterpret_cast<float4*>(s_act)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // LDS stores of the weights: layer 0 now (its loads were issued first and return first); the other layers after layer
  // 0 has been published, inside the wait for the first exchange -- the prologue ends when the FIRST loads are back, not
  // the last (the compiler's vmcnt waits follow the issue order)
  auto store_layer = [&](int l, int woffs) {
    const bool in_x = get_layer(a, l).in_x != 0;
    const int hp = l ? H : 0, WP = hp + (in_x ? XW : 0) + 4;
    if (tid < NC) s_bias[l * NC + tid] = bv[l];
    if (l > 0) *reinterpret_cast<float4*>(s_w + woffs + ch * WP + kh) = vh[l];
    if (in_x && tid < NT / 2 && kx < XW) *reinterpret_cast<float4*>(s_w + woffs + cx * WP + hp + kx) = vx[l];
    return NC * WP;
  };
  const int w_after0 = store_layer(0, 0);
  if (tid == 0) s_misc[0] = cnt, s_misc[1] = 0, s_misc[2] = stamps_on;
  __syncthreads();
  const unsigned count = s_misc[0];
  const size_t img_floats = (size_t) nX * G * Bp * NC;
  float* img = a.exch + (count & 1u) * img_floats;
  repoison(a.exch + ((count & 1u) ^ 1u) * img_floats, nX, G, Bp, g, a.census + ((count & 1u) ^ 1u) * (G_NET * 4));
  census_publish(a, count & 1u, g);
  const unsigned* census = a.census + (count & 1u) * (G_NET * 4);
  bool plain = false;  // slabs leave as plain stores once the census has shown the whole network on one XCD (xcd_mode 1 / 3)
  unsigned census_word = 0;
  int si = 0;
  stamp(a, s_misc, si, t_entry);

  int woff = 0;
  for (int l = 0; l < nL; ++l) {
    const FusedLayer L = get_layer(a, l);
    const int hp = l ? H : 0, WP = hp + (L.in_x ? XW : 0) + 4;
    const bool last = l == nL - 1;
    if (l > 0) {
      const bool ok = gather_slabs<U>(img + (size_t) (l - 1) * G * Bp * NC, s_act, B, Bp, HP, B * H / 4, census, census_word);
      if (!ok) s_misc[1] = 1;
      __syncthreads();
      if (s_misc[1]) break;
      if (l == 1) plain = (a.xcd_mode == 1 || a.xcd_mode == 3) && census_one_xcd(census_word);
      stamp(a, s_misc, si);
    }
    if (last && col0 >= L.out) break;
    f4 acc[ROUNDS][2];
    zero_acc<ROUNDS>(acc);
    if (l > 0) mfma_dot<ROUNDS, H / NW / 4>(acc, s_act, HP, rowc, s_w + woff, WP, lm.colw, (H / NW) * wave);
    if (L.in_x) mfma_dot<ROUNDS>(acc, s_x0, XP, rowc, s_w + woff + hp, WP, lm.colw, (XW / NW) * wave, XW / NW / 4);
    write_partials<ROUNDS>(s_part, acc, lm, wave, Bp, 4 * PASSES);
    __syncthreads();  // partials complete; s_act / s_x0 reads of this layer done
    if (tid < Bp * 2) {  // one 16-byte unit per thread: bias, activation, the "never the sentinel" rule; the slab leaves as whole lines
      const int row = tid >> 1, part = tid & 1;
      float4 y = sum_partials(s_part, Bp, row, part);
      const float4 bz = *reinterpret_cast<const float4*>(s_bias + l * NC + 4 * part);
      y.x += bz.x, y.y += bz.y, y.z += bz.z, y.w += bz.w;
      if (L.relu) y.x = fmaxf(y.x, 0.f), y.y = fmaxf(y.y, 0.f), y.z = fmaxf(y.z, 0.f), y.w = fmaxf(y.w, 0.f);
      y.x = not_sentinel(y.x), y.y = not_sentinel(y.y), y.z = not_sentinel(y.z), y.w = not_sentinel(y.w);
      if (row < B) {
        if (last) {
          const int c = col0 + 4 * part;
          if (c == 0) *reinterpret_cast<float4*>(s_raw + 4 * row) = y;
          if (c < L.out) *const_cast<float*>(head_elem_sel(a, a.head_out, a.out, row, c, L.out)) = y.x;
          if (c + 1 < L.out) *const_cast<float*>(head_elem_sel(a, a.head_out, a.out, row, c + 1, L.out)) = y.y;
          if (c + 2 < L.out) *const_cast<float*>(head_elem_sel(a, a.head_out, a.out, row, c + 2, L.out)) = y.z;
          if (c + 3 < L.out) *const_cast<float*>(head_elem_sel(a, a.head_out, a.out, row, c + 3, L.out)) = y.w;
          if (a.sk_cache) {  // (no_grad copy for test-time interpolation, sk_gs.py:1077-1085)
            const int frame = frame_pf;  // (fetched in the prologue: here it was a round trip in front of the cache row's stores)
            float* cr = a.sk_cache + ((size_t) frame * B + row) * L.out;
            if (c == 0) {  // F.normalize(raw + [0, 0, 0, 1]): the expression of chain::stage_skeleton
              const chain::Q4 q = chain::qnormalize({y.x, y.y, y.z, y.w + 1.0f});
              cr[0] = q.x, cr[1] = q.y, cr[2] = q.z, cr[3] = q.w;
            } else {
              if (c < L.out) cr[c] = y.x;
              if (c + 1 < L.out) cr[c + 1] = y.y;
              if (c + 2 < L.out) cr[c + 2] = y.z;
              if (c + 3 < L.out) cr[c + 3] = y.w;
            }
          }
        } else {
          store16_exch(img + ((size_t) l * G + g) * Bp * NC + 4 * tid, y, plain);
          *reinterpret_cast<float4*>(a.acts + ((size_t) l * B + row) * H + col0 + 4 * part) = y;
        }
      }
    }
    woff += NC * WP;
    if (l == 0) {  // (made visible by the barrier behind the next gather)
      int woffs = w_after0;
#pragma unroll
      for (int l2 = 1; l2 < KL; ++l2)
        if (l2 < nL) woffs += store_layer(l2, woffs);
    }
    stamp(a, s_misc, si);
  }
  if (a.has_chain && g == 0 && !s_misc[1]) {  // joint rotations -> bone transforms (bone_chain.inl), s_part as scratch
    __syncthreads();
    chain::ChainArgs c = a.chain;
    c.sk_r_raw = s_raw;
    chain::forward_body(s_part, c, cpf);
  }
  // the optional copy of the encoded input leaves from the last workgroup, after its part of the chain
  if (a.x0 && g == G - 1)
    for (int i = tid; i < B * IN; i += NT) a.x0[i] = s_x0[(i / IN) * XP + (i % IN)];
  if (g == 0 && tid == 0) {
    gu32* h = reinterpret_cast<gu32*>((unsigned long long) a.hdr);
    if (s_misc[1]) __hip_atomic_fetch_add(h + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (plain) __hip_atomic_fetch_add(h + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // launches whose exchange stayed in one L2
    __hip_atomic_fetch_or(h + 6, 1u << (xcc_id() & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the XCDs workgroup 0 has run on
    __hip_atomic_store(h, count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// --------------------------------------------------------------------------------------------------------- backward
// LDS (floats): s_gz [Bp][HP] (later reused for the encoded input, pitch INP) | s_own [n_layers][Bp][NC] (this workgroup's
//               slab of every gZ_l, kept for the weight gradients) | s_part [NW][Bp][NC] | transposed slabs: layer l >= 1 ->
//               T_l [NC][op_l + 4] = W_l[o][col0 + c], op_l = out_l rounded up to 32; with g_x0 and col0 < IN also X_l
//               [NC][op_l + 4] = W_l[o][in_h + col0 + c] for every layer with in_x > 0 | misc
template <int PASSES>
__global__ void __launch_bounds__(NT) fused_mlp_backward_kernel(const FusedArgs a) {
  constexpr int Bp = 16 * PASSES, ROUNDS = (4 * PASSES + 7) / 8;
  constexpr int U = (Bp * 64 + NT - 1) / NT <= 4 ? 4 : 8;
  static_assert((Bp * 64 + NT - 1) / NT <= 8, "gather_slabs covers at most 8 units per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Role role = role_of(a);
  if (!role.net) {
    adam_side_job(a, role.wg, role.n_side);
    return;
  }
  const int tid = threadIdx.x, wave = tid >> 6;
  const int G = G_NET, g = role.g, col0 = g * NC;
  const int B = a.B, nL = a.n_layers, IN = a.IN, nX = nL - 1;
  const bool want_gx = a.g_x0 != nullptr && col0 < IN;
  float* s_gz   = smem;
  float* s_x0   = s_gz;  // the encoded input is only needed by the weight gradients: built after the chain, over s_gz
  float* s_own  = s_gz + Bp * HP;
  float* s_part = s_own + nL * Bp * NC;
  float* s_graw = s_part + NW * Bp * NC;  // [Bp][4]: gradient of the raw joint rotations (head 0) from the chain backward
  float* s_wT   = s_graw + Bp * 4;
  int t_total = 0;  // floats of the T slabs; the X slabs follow
  for (int l = 1; l < nL; ++l) t_total += NC * (pad32(get_layer(a, l).out) + 4);
  int x_total = 0;
  if (want_gx)
    for (int l = 0; l < nL; ++l)
      if (get_layer(a, l).in_x) x_total += NC * (pad32(get_layer(a, l).out) + 4);
  unsigned* s_misc = reinterpret_cast<unsigned*>(smem + a.lds_floats - 4);
  // per-layer data the hop loop reads at a RUN-TIME layer index (the loop is rolled: one copy of its code, warm in the instruction
  // cache after the first hop): the gW / gb pointers (a dynamic index into the kernel arguments is a scalar load from the
  // host-visible argument segment); the ReLU masks of this workgroup's slab are four bits per layer in a register pair
  float** s_ptr = reinterpret_cast<float**>(smem + a.lds_floats - 4 - 4 * KL);   // [KL][2]: gW_l, gb_l
  static_assert(4 * (KL - 1) <= 64, "the ReLU masks of a thread's unit: four bits per layer in one 64-bit word");
  unsigned long long relu_bits = 0;  // bit 4 l + k: element k of this thread's unit of a_l is > 0 (or layer l has no ReLU)
  const LaneMap lm = lane_map();
  int rowc[ROUNDS];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) rowc[rd] = (8 * rd + lm.rgl < 4 * PASSES) ? 4 * (8 * rd + lm.rgl) + lm.j : lm.j;

  // ---- prologue: one memory round trip (see the forward kernel)
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
  unsigned cnt = 0, stamps_on = 0;
  if (tid == 0) {
    const gu32* h = reinterpret_cast<const gu32*>((unsigned long long) a.hdr);
    cnt       = __hip_atomic_load(h + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamps_on = __hip_atomic_load(h + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // diagnostics: stamps 18.. of workgroup 0 = entry, chain inputs staged, weight slabs in LDS, chain levels walked
  auto pstamp = [&](int slot, unsigned long long t) {
    if (stamps_on && blockIdx.x == 0 && tid == 0) {
      a.hdr[16 + 2 * slot]     = (unsigned) (t ? t : __builtin_amdgcn_s_memrealtime());
      a.hdr[16 + 2 * slot + 1] = (unsigned) __builtin_amdgcn_s_memtime();
    }
  };
  // EVERY global load of the prologue goes out now, in one batch: the chain backward's inputs (the frame's row of global_T from the
  // copy the forward left in the workspace header -- read through frame_index[0] it was a second, dependent round trip), the heads'
  // incoming gradient, the encoded input, the weights.  (Round 6, stamps of workgroup 0 beside the Adam stream: index -> row, then the
  // weights' round trip, then the tree walk, then the head gradients' round trip, one after the other, were 15.8 us of prologue.  The
  // walk cannot be made to overlap the weights' round trip from C++: hipcc's wait counts turn into vmcnt(0) at its first branch.)
  chain::Prefetch cpf;
  cpf.valid = false;
  if (a.has_chain) cpf = chain::prefetch(a.chain, true, true);
  pstamp(18, t_entry);
  // the ReLU masks of this workgroup's slab, one 16-byte unit per thread and layer (threads < 2 Bp)
  const int mrow = tid >> 1, mpart = tid & 1;
  const bool mlive = tid < Bp * 2 && mrow < B;
  float4 am[KL], vt[KL], vx[KL];
  // the encoded input (needed after the chain, by the weight gradients of the layers that read it): the forward's copy,
  // every load unconditional (clamped index) and in flight with the rest of the prologue
  constexpr int XQ = Bp * INP / NT;
  float xv[XQ];
  if (a.x0) {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int i = tid + q * NT, b = i / INP, c = i - b * INP;
      xv[q] = a.x0[(size_t) min(b, B - 1) * IN + min(c, IN - 1)];
    }
  }
  const int wo = tid >> 1, wc4 = 4 * (tid & 1);  // (weight row o, group of four columns): NT units
  {
    const float* wp[KL];  // (see the forward prologue: all pointer loads up front, one wait)
    float* gwp[KL];
    float* gbp[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) wp[l] = a.W[l], gwp[l] = a.gW[l], gbp[l] = a.gb[l];
#pragma unroll
    for (int l = 0; l < KL; ++l) asm volatile("" : "+s"(wp[l]), "+s"(gwp[l]), "+s"(gbp[l]));
    if (tid == 0) {
#pragma unroll
      for (int l = 0; l < KL; ++l) s_ptr[2 * l] = gwp[l], s_ptr[2 * l + 1] = gbp[l];
    }
#pragma unroll
    for (int l = 0; l < KL; ++l) {
      am[l] = make_float4(1.f, 1.f, 1.f, 1.f);
      vt[l] = vx[l] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l < nL) {
        FusedLayer L = get_layer(a, l);
        L.W = wp[l];
        const int K = L.in_h + L.in_x;
        if (l < nL - 1 && L.relu && mlive)
          am[l] = *reinterpret_cast<const float4*>(a.acts + ((size_t) l * B + mrow) * H + col0 + 4 * mpart);
        if (wo < L.out) {
          if (l >= 1) vt[l] = *reinterpret_cast<const float4*>(L.W + (size_t) wo * K + col0 + wc4);
          if (want_gx && L.in_x && col0 + wc4 < L.in_x) vx[l] = *reinterpret_cast<const float4*>(L.W + (size_t) wo * K + L.in_h + col0 + wc4);
        }
      }
    }
  }
  // the incoming gradient of the heads (gZ of the last layer) the chain does not produce: this thread's entries of the [Bp][64] image
  // and of this workgroup's slab, fetched in the same batch, behind the weights (instead of behind the walk)
  const int oL_pf = get_layer(a, nL - 1).out;
  constexpr int GZQ = Bp * 64 / NT;
  float hgz[GZQ], hown = 0.f;
  {
    const int c = tid & 63;
#pragma unroll
    for (int q = 0; q < GZQ; ++q) {
      const int b = (tid >> 6) + q * (NT / 64);
      hgz[q] = 0.f;
      if (b < B && c < oL_pf && !(a.has_chain && c < 4))
        hgz[q] = *head_elem_sel(a, a.head_gout, a.g_out, b, c, oL_pf);
    }
    const int bo = tid / NC, co = col0 + tid % NC;
    if (tid < Bp * NC && bo < B && co < oL_pf && !(a.has_chain && co < 4))
      hown = *head_elem_sel(a, a.head_gout, a.g_out, bo, co, oL_pf);
  }
  if (stamps_on_all(a)) {  // (diagnostics only: when every load of the batch is back)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pstamp(22, 0);
  }
  if (a.has_chain) {  // (LDS only)
    chain::backward_stage(s_part, a.chain, cpf);
    pstamp(19, 0);
    chain::backward_levels(s_part, a.chain, s_graw, g == 0);
    __syncthreads();
  }
  pstamp(21, 0);
  {
    int toff = 0, xoff = t_total;
#pragma unroll
    for (int l = 0; l < KL; ++l) {
      if (l < nL) {
        if (l < nL - 1)
          relu_bits |= (unsigned long long) ((am[l].x > 0.f ? 1u : 0u) | (am[l].y > 0.f ? 2u : 0u) | (am[l].z > 0.f ? 4u : 0u) | (am[l].w > 0.f ? 8u : 0u))
                       << (4 * l);
        const int op = pad32(get_layer(a, l).out), OP = op + 4;
        if (wo < op) {  // rows out_l .. op_l - 1 are written as zeros
          if (l >= 1) {
            float* d = s_wT + toff + wc4 * OP + wo;
            d[0] = vt[l].x, d[OP] = vt[l].y, d[2 * OP] = vt[l].z, d[3 * OP] = vt[l].w;
          }
          if (want_gx && get_layer(a, l).in_x) {
            float* d = s_wT + xoff + wc4 * OP + wo;
            d[0] = vx[l].x, d[OP] = vx[l].y, d[2 * OP] = vx[l].z, d[3 * OP] = vx[l].w;
          }
        }
        if (l >= 1) toff += NC * OP;
        if (want_gx && get_layer(a, l).in_x) xoff += NC * OP;
      }
    }
  }
  pstamp(20, 0);
  {  // gZ of the last layer = the incoming gradient (zero-padded to 64 columns); this workgroup's slab of it: the chain's part from
     // LDS, the rest from the registers filled at the top
    const int oL = oL_pf;
    const int c = tid & 63;
#pragma unroll
    for (int q = 0; q < GZQ; ++q) {
      const int b = (tid >> 6) + q * (NT / 64);
      float v = hgz[q];
      if (a.has_chain && c < 4 && b < B) v = s_graw[4 * b + c];
      s_gz[b * HP + c] = (b < B && c < oL) ? v : 0.f;
    }
    if (tid < Bp * NC) {
      const int b = tid / NC, cc = col0 + tid % NC;
      float v = hown;
      if (a.has_chain && cc < 4 && b < B) v = s_graw[4 * b + cc];
      s_own[(nL - 1) * Bp * NC + tid] = (b < B && cc < oL) ? v : 0.f;
    }
  }
  if (tid == 0) s_misc[0] = cnt, s_misc[1] = 0, s_misc[2] = stamps_on;
  __syncthreads();
  const unsigned count = s_misc[0];
  const size_t img_floats = (size_t) nX * G * Bp * NC;
  float* img = a.exch + (count & 1u) * img_floats;
  repoison(a.exch + ((count & 1u) ^ 1u) * img_floats, nX, G, Bp, g, a.census + ((count & 1u) ^ 1u) * (G_NET * 4));
  census_publish(a, count & 1u, g);
  const unsigned* census = a.census + (count & 1u) * (G_NET * 4);
  bool plain = false;  // slabs leave as plain stores once the census has shown the whole network on one XCD (xcd_mode 1 / 3)
  unsigned census_word = 0;
  int si = 12;  // diagnostics: stamps 12.. = prologue done, chain done, input gradient done, weight gradients done
  stamp(a, s_misc, si);

  const int kcol = (H / NW) * wave + 4 * lm.rgl + lm.j;  // weight-gradient column of this lane (block = 2 k-group + cg)
  f4 gw = {0.f, 0.f, 0.f, 0.f};  // weight-gradient block of the layer in flight, its destination, row pitch and live rows
  float* gw_dst = nullptr;
  int gw_ld = 0, gw_rows = 0;
  f4 accx[ROUNDS][2];  // the input gradient of this workgroup's columns accumulates over the layers that read x0
  zero_acc<ROUNDS>(accx);

  // ---- (A) the dependent chain: gA_{l-1}[:, slab] = gZ_l W_l[:, slab], masked by the ReLU of layer l-1.  The slabs were
  // laid out by increasing layer: walk their offsets backwards.
  int tcur = t_total, xcur = t_total + x_total;
#pragma unroll 1
  for (int l = nL - 1; l >= 1; --l) {  // ROLLED: every hop runs the same instructions (unrolled nine times the body was ~130 KB of code,
    {                                   // each hop's copy fetched cold; the per-layer data comes from LDS / masks instead of kernel arguments)
      const int L_out = l == nL - 1 ? a.out_last : H, L_in_x = ((a.xmask >> l) & 1u) ? a.IN : 0, L_in_h = H;
      float* const L_gW = s_ptr[2 * l];
      const int op = pad32(L_out), OP = op + 4;
      const bool publish = l - 1 >= 1 || a.g_x0 != nullptr;  // gZ_0 is only exchanged for the input gradient
      const bool do_x = want_gx && L_in_x;
      tcur -= NC * OP;
      if (do_x) xcur -= NC * OP;
      if (gw_dst) {  // the previous iteration's weight-gradient block
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < gw_rows) gw_dst[(size_t) i * gw_ld] = gw[i];
        gw_dst = nullptr;
      }
      // (B, hidden part) this workgroup's rows of gW_l = gZ_l[:, slab]^T a_{l-1}: the activation column of this lane, one
      // load per row, all in flight behind the chain product below; the outer products run after the publish, while the
      // other workgroups' slabs are on their way (nothing on the chain waits for them).  (Loading them one iteration
      // ahead, into a second register set, was slower: 30.6 -> 32.2 us.)
      const bool wg_rows = col0 < L_out;
      float av[Bp];
      if (wg_rows) load_act_column<Bp>(av, a.acts + (size_t) (l - 1) * B * H + kcol, B);
      f4 acc[ROUNDS][2];
      zero_acc<ROUNDS>(acc);
      if (op == H) {
        mfma_dot<ROUNDS, H / NW / 4>(acc, s_gz, HP, rowc, s_wT + tcur, OP, lm.colw, (H / NW) * wave);
        if (do_x) mfma_dot<ROUNDS, H / NW / 4>(accx, s_gz, HP, rowc, s_wT + xcur, OP, lm.colw, (H / NW) * wave);
      } else {
        mfma_dot<ROUNDS>(acc, s_gz, HP, rowc, s_wT + tcur, OP, lm.colw, (op / NW) * wave, op / NW / 4);
        if (do_x) mfma_dot<ROUNDS>(accx, s_gz, HP, rowc, s_wT + xcur, OP, lm.colw, (op / NW) * wave, op / NW / 4);
      }
      write_partials<ROUNDS>(s_part, acc, lm, wave, Bp, 4 * PASSES);
      __syncthreads();  // partials complete; s_gz has been read
      if (tid < Bp * 2) {
        float4 y = sum_partials(s_part, Bp, mrow, mpart);
        const unsigned m = (unsigned) (relu_bits >> (4 * (l - 1)));
        y.x = (mlive && (m & 1u)) ? not_sentinel(y.x) : 0.f, y.y = (mlive && (m & 2u)) ? not_sentinel(y.y) : 0.f;
        y.z = (mlive && (m & 4u)) ? not_sentinel(y.z) : 0.f, y.w = (mlive && (m & 8u)) ? not_sentinel(y.w) : 0.f;
        if (publish && mlive) store16_exch(img + ((size_t) (l - 1) * G + g) * Bp * NC + 4 * tid, y, plain);
        *reinterpret_cast<float4*>(s_own + (l - 1) * Bp * NC + 4 * tid) = y;  // kept for the weight gradients
      }
      if (wg_rows) {
        const float* own = s_own + l * Bp * NC + lm.colw;  // A operand: gZ_l[b][4 cg + j]; B operand: a_{l-1}[b][kcol]
        gw = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b4 = 0; b4 < Bp; b4 += 4) {
          if (b4 < B) {
#pragma unroll
            for (int u = 0; u < 4; ++u) gw = __builtin_amdgcn_mfma_f32_4x4x1f32(own[(b4 + u) * NC], av[b4 + u], gw, 0, 0, 0);
          }
        }
        // (stored at the top of the next iteration: stores still in flight here would be waited for by the gather's
        // s_waitcnt vmcnt(0) together with its loads)
        gw_dst = L_gW + (size_t) (col0 + 4 * lm.cg) * (L_in_h + L_in_x) + kcol;  // D: VGPR i = row 4 cg + i of the slab
        gw_ld = L_in_h + L_in_x, gw_rows = L_out - (col0 + 4 * lm.cg);
      }
      if (publish) {
        const bool ok = gather_slabs<U>(img + (size_t) (l - 1) * G * Bp * NC, s_gz, B, Bp, HP, B * H / 4, census, census_word);
        if (!ok) s_misc[1] = 1;
      }
      __syncthreads();
      if (s_misc[1]) break;
      if (publish && l == nL - 1) plain = (a.xcd_mode == 1 || a.xcd_mode == 3) && census_one_xcd(census_word);
    }
  }
  if (gw_dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < gw_rows) gw_dst[(size_t) i * gw_ld] = gw[i];
  }
  stamp(a, s_misc, si);
  // ---- input gradient: the layers that read x0 (layer 0 included) contribute gZ_l W_l[:, x0 part]
  if (want_gx && !s_misc[1]) {
    mfma_dot<ROUNDS, H / NW / 4>(accx, s_gz, HP, rowc, s_wT + t_total, H + 4, lm.colw, (H / NW) * wave);  // X_0 (out = H): the first x0 slab
    write_partials<ROUNDS>(s_part, accx, lm, wave, Bp, 4 * PASSES);
    __syncthreads();
    if (tid < Bp * 2 && mrow < B) {
      const float4 y = sum_partials(s_part, Bp, mrow, mpart);
      const int c = col0 + 4 * mpart;
      float* d = a.g_x0 + (size_t) mrow * IN + c;
      if (c < IN) d[0] = y.x;
      if (c + 1 < IN) d[1] = y.y;
      if (c + 2 < IN) d[2] = y.z;
      if (c + 3 < IN) d[3] = y.w;
    }
  }
  stamp(a, s_misc, si);
  // ---- (B, the rest) after the chain: the encoded-input columns of the weight gradients (layer 0 and the skip layers) and
  // the bias gradients, from the kept slabs gZ_l[:, slab]
  if (!s_misc[1]) {
    __syncthreads();  // the last readers of s_gz are done: it becomes the encoded input
    if (a.x0) {  // the forward's copy, fetched in the prologue
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int i = tid + q * NT, b = i / INP, c = i - b * INP;
        s_x0[i] = (b < B && c < IN) ? xv[q] : 0.f;
      }
    } else {
      const int pe = a.p_dim * (1 + 2 * a.p_deg);
      for (int i = tid; i < Bp * INP; i += NT) {
        const int b = i / INP, c = i - b * INP;
        float v = 0.f;
        if (b < B && c < IN) {
          const int cc = c < pe ? c : c - pe, D = c < pe ? a.p_dim : a.t_dim;
          v = c < pe ? a.points[(size_t) b * a.p_dim + cc % D] : a.t[cc % D];
          if (cc >= D) {
            const int col = cc / D - 1;
            v = sinf(scalbnf(v, col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
          }
        }
        s_x0[i] = v;
      }
    }
    __syncthreads();
    for (int l = nL - 1; l >= 0; --l) {
      struct { int out, in_x, in_h; float* gW; } L = {l == nL - 1 ? a.out_last : H, ((a.xmask >> l) & 1u) ? a.IN : 0, l ? H : 0, s_ptr[2 * l]};
      if (col0 >= L.out || !L.in_x) continue;
      const int K = L.in_h + L.in_x;
      const float* own = s_own + l * Bp * NC;
      for (int i = tid; i < L.in_x * (NC / 4); i += NT) {  // thread <-> (x0 column of gW_l, group of four rows)
        const int kx = i % L.in_x, q = i / L.in_x, k = L.in_h + kx;
        float gacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
          const float av = s_x0[b * INP + kx];
          const float4 o = *reinterpret_cast<const float4*>(own + b * NC + 4 * q);
          gacc[0] += o.x * av, gacc[1] += o.y * av, gacc[2] += o.z * av, gacc[3] += o.w * av;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (col0 + 4 * q + c < L.out) L.gW[(size_t) (col0 + 4 * q + c) * K + k] = gacc[c];
      }
    }
    // bias gradients: one thread per (layer, column) -- as a loop over the layers in eight lanes this was nine serial
    // chains of B dependent LDS reads, 5 us at the end of the launch
    for (int i = tid; i < nL * NC; i += NT) {
      const int l = i / NC, c = i - l * NC;
      struct { int out; float* gb; } L = {l == nL - 1 ? a.out_last : H, s_ptr[2 * l + 1]};
      if (L.gb && col0 + c < L.out) {
        const float* own = s_own + l * Bp * NC + c;
        float sum = 0.f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) sum += own[b * NC];
        L.gb[col0 + c] = sum;
      }
    }
  }
  stamp(a, s_misc, si);
  if (g == 0 && tid == 0) {
    gu32* h = reinterpret_cast<gu32*>((unsigned long long) a.hdr);
    if (s_misc[1]) __hip_atomic_fetch_add(h + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (plain) __hip_atomic_fetch_add(h + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_or(h + 7, 1u << (xcc_id() & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(h + 3, count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------ host
constexpr int HDR_BYTES = 256;
constexpr int CENSUS_BYTES = 2 * G_NET * 16;  // one direction: both parities of [G][4 words]

struct Plan {
  int Bp, passes, IN, INP, G, NC;
  size_t exch_bytes;  // one direction: both parities
};
int make_plan(const skgs_mlp_desc* d, Plan* p) {
  SKGS_REQUIRE(d, "deform_mlp: NULL descriptor");
  SKGS_REQUIRE(d->B >= 1 && d->B <= 48, "deform_mlp: the fused kernels handle 1..48 rows (B = %d): use the per-layer path", d->B);
  SKGS_REQUIRE(d->hidden == H, "deform_mlp: the fused kernels are built for hidden width %d (got %d): use the per-layer path", H, d->hidden);
  SKGS_REQUIRE(d->n_layers >= 2 && d->n_layers <= KL, "deform_mlp: 2..%d layers (heads included)", KL);
  SKGS_REQUIRE(d->p_dim >= 1 && d->t_dim >= 0 && d->p_degree >= 0 && d->t_degree >= 0, "deform_mlp: bad encoder sizes");
  p->IN  = d->p_dim * (1 + 2 * d->p_degree) + d->t_dim * (1 + 2 * d->t_degree);
  p->INP = (p->IN + 63) & ~63;
  SKGS_REQUIRE(p->IN <= INP && p->IN % 4 == 0, "deform_mlp: encoded input must be a multiple of 4 and <= %d wide (%d)", INP, p->IN);
  for (int l = 0; l < d->n_layers; ++l) {
    const skgs_mlp_layer& L = d->layer[l];
    SKGS_REQUIRE(L.W && (reinterpret_cast<uintptr_t>(L.W) & 15) == 0, "deform_mlp: layer %d: weights missing or not 16-byte aligned", l);
    SKGS_REQUIRE(L.in_hidden == (l == 0 ? 0 : d->hidden), "deform_mlp: layer %d: in_hidden must be %d", l, l == 0 ? 0 : d->hidden);
    SKGS_REQUIRE(L.in_x0 == 0 || L.in_x0 == p->IN, "deform_mlp: layer %d: in_x0 must be 0 or the encoded width %d", l, p->IN);
    SKGS_REQUIRE(l > 0 || L.in_x0 == p->IN, "deform_mlp: the first layer reads the encoded input");
    SKGS_REQUIRE(l == d->n_layers - 1 ? (L.out >= 1 && L.out <= d->hidden) : L.out == d->hidden,
        "deform_mlp: layer %d: out = %d", l, L.out);
  }
  SKGS_REQUIRE(d->n_heads >= 0 && d->n_heads <= 4, "deform_mlp: 0..4 heads");
  if (d->n_heads) {
    int sum = 0;
    for (int j = 0; j < d->n_heads; ++j) sum += d->head_dim[j];
    SKGS_REQUIRE(sum == d->layer[d->n_layers - 1].out, "deform_mlp: the head widths must add up to the last layer's out");
  }
  p->passes = (d->B + 15) / 16;
  p->Bp     = p->passes * 16;
  p->NC     = NC;
  p->G      = d->hidden / NC;
  p->exch_bytes = (size_t) 2 * (d->n_layers - 1) * p->Bp * d->hidden * 4 + CENSUS_BYTES;  // images, then the census
  return 0;
}

void fill_args(const skgs_mlp_desc* d, const Plan& p, FusedArgs* a) {
  a->B = d->B, a->p_dim = d->p_dim, a->p_deg = d->p_degree, a->t_dim = d->t_dim, a->t_deg = d->t_degree;
  a->IN = p.IN, a->n_layers = d->n_layers;
  a->xmask = a->relu_mask = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const skgs_mlp_layer& L = d->layer[l];
    a->W[l] = L.W, a->bias[l] = L.bias, a->gW[l] = L.gW, a->gb[l] = L.gb;
    if (L.in_x0) a->xmask |= 1u << l;
    if (L.relu) a->relu_mask |= 1u << l;
  }
  a->out_last = d->layer[d->n_layers - 1].out;
  a->n_heads = d->n_heads;
  for (int j = 0; j < 4; ++j) a->head_dim[j] = d->head_dim[j], a->head_out[j] = d->head_out[j], a->head_gout[j] = d->head_gout[j];
}

// the kinematic chain riding on a launch: the network's rows are the bones, its first head their raw rotations
int fill_chain(const skgs_mlp_desc* d, const Plan& p, const skgs_bone_chain_desc* b, FusedArgs* a) {
  SKGS_REQUIRE(d->n_heads >= 1 && d->head_dim[0] == 4, "skeleton: the network's first head must be the [M,4] raw joint rotations");
  SKGS_REQUIRE(b->M == d->B, "skeleton: %d bones but %d network rows", b->M, d->B);
  SKGS_REQUIRE(b->root >= 0 && b->root < b->M && b->num_levels >= 1, "skeleton: bad skeleton sizes");
  SKGS_REQUIRE(b->parents && b->level_nodes && b->level_start && b->joints, "skeleton: NULL topology / joints");
  a->has_chain = 1;
  chain::ChainArgs& c = a->chain;
  c.M = b->M, c.root = b->root, c.num_levels = b->num_levels, c.parents = b->parents, c.level_nodes = b->level_nodes;
  c.level_start = b->level_start, c.joints = b->joints, c.global_T = b->global_T, c.frame_index = b->frame_index;
  c.bone_T = b->bone_T, c.chain_A = b->chain_A;
  c.global_T_row = reinterpret_cast<float*>(a->hdr + 8);  // header words 8..14: the frame's row, forward -> backward (a.hdr is set before)
  a->sk_cache = b->sk_cache;
  return 0;
}
int fill_side(const skgs_adam_range* side, FusedArgs* a) {
  if (!side || side->n_tensors <= 0) return 0;
  SKGS_REQUIRE(side->tensors && side->step_count && side->chunk_begin >= 0 && side->chunk_end >= side->chunk_begin,
      "deform_mlp: bad side range");
  a->adam_tensors = reinterpret_cast<const AdamTensor*>(side->tensors), a->adam_n = side->n_tensors;
  a->adam_c0 = side->chunk_begin, a->adam_c1 = side->chunk_end;
  a->adam_beta1 = side->beta1, a->adam_beta2 = side->beta2, a->adam_eps = (float) side->eps;
  a->adam_step  = reinterpret_cast<const AdamState*>(side->step_count);
  a->adam_after_advance = side->after_advance ? 1 : 0;
  return 0;
}
int forward_impl(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t, float* x0,
    float* acts, float* out, void* workspace, size_t workspace_bytes, const skgs_adam_range* side, skgs_stream_t stream);
int backward_impl(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    const float* x0, const float* acts, const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes,
    const skgs_adam_range* side, skgs_stream_t stream);

// Mode 1 puts all 32 network workgroups of a launch on ONE XCD, one per CU (the launch's LDS request): TWO such launches dispatched
// within the same microsecond to the same XCD interleave on its 32 CUs, and each then holds CUs the other's missing workgroups
// need -- both give up after their bounded spins (a loud failure, never a wrong result).  One launch at a time per device is what a
// training process does; ranks that SHARE a GPU (SKGS_SHARE_GPU=1: the tests' stand-in for a multi-GPU node) run in lockstep and do
// hit that window (1 of 3 sessions, 2 ranks): they get mode 0, where the network's workgroups are the first blocks of their
// launch, four per XCD, and eight launches fit side by side.
int g_xcd_mode = -1;  // -1: not decided yet (the environment's SKGS_MLP_XCD; default 1, or 0 with SKGS_SHARE_GPU=1)
inline int xcd_mode_wanted() {
  if (g_xcd_mode < 0) {
    const char* e = getenv("SKGS_MLP_XCD");
    const char* shared = getenv("SKGS_SHARE_GPU");
    g_xcd_mode = e ? std::max(0, std::min(3, atoi(e))) : ((shared && atoi(shared) != 0) ? 0 : 1);
  }
  return g_xcd_mode;
}
template <typename KernelT>
int launch(KernelT k, const Plan& p, const FusedArgs& a, size_t lds, hipStream_t s, int prof_id) {
  ProfScope prof(prof_id, s);
  if (lds > 64 * 1024)
    SKGS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // side job: one workgroup per CU the network leaves idle (each takes two chunks per iteration)
  int side = 0;
  if (a.adam_tensors && a.adam_c1 > a.adam_c0)
    side = (int) std::max<long long>(1, std::min<long long>((a.adam_c1 - a.adam_c0 + 1) / 2, num_cus() - p.G));  // never 0: the rows MUST be updated
  FusedArgs b = a;
  int grid = p.G + side;
  {
    static const int delay = [] { const char* e = getenv("SKGS_SIDE_DELAY"); return e ? std::max(0, atoi(e)) : 0; }();
    b.side_delay = delay;
  }
  b.xcd_mode = xcd_mode_wanted();
  if (b.xcd_mode) {  // the network on blocks 0, 8, .. 8 (G - 1): the grid must reach the last of them
    if (num_cus() != 256 || p.G != G_NET || (side > 0 && grid < 8 * p.G - 7)) b.xcd_mode = 0;
    else grid = std::max(grid, 8 * p.G - 7);
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, s, b);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
#define SKGS_MLP_DISPATCH(KERNEL, PROF_ID)                                                    \
  if (p.passes == 1) return launch(KERNEL<1>, p, a, lds, (hipStream_t) stream, PROF_ID);      \
  if (p.passes == 2) return launch(KERNEL<2>, p, a, lds, (hipStream_t) stream, PROF_ID);      \
  return launch(KERNEL<3>, p, a, lds, (hipStream_t) stream, PROF_ID);

__global__ void init_workspace_kernel(uint32_t* w, size_t n_words) {
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
    w[i] = i < HDR_BYTES / 4 ? 0u : SENTINEL;
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

int32_t skgs_deform_mlp_xcd_mode(int32_t mode) {
  const int before = xcd_mode_wanted();
  if (mode >= 0) g_xcd_mode = std::min(3, (int) mode);
  return before;
}

size_t skgs_deform_mlp_workspace_bytes(const skgs_mlp_desc* d) {
  Plan p;
  if (make_plan(d, &p)) return 0;
  return HDR_BYTES + 2 * p.exch_bytes;
}

int skgs_deform_mlp_workspace_init(void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  SKGS_REQUIRE(workspace && workspace_bytes >= HDR_BYTES && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
      "deform_mlp_workspace_init: NULL, short or unaligned workspace");
  const size_t n = workspace_bytes / 4;
  hipLaunchKernelGGL(init_workspace_kernel, dim3((unsigned) std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0,
      (hipStream_t) stream, reinterpret_cast<uint32_t*>(workspace), n);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_deform_mlp_forward(const skgs_mlp_desc* d, const float* points, const float* t, float* x0, float* acts, float* out,
    void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  return forward_impl(d, nullptr, points, t, x0, acts, out, workspace, workspace_bytes, nullptr, stream);
}

int skgs_skeleton_forward(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    float* x0, float* acts, void* workspace, size_t workspace_bytes, const skgs_adam_range* side, skgs_stream_t stream) {
  SKGS_REQUIRE(bones, "skeleton_forward: NULL bone chain");
  return forward_impl(d, bones, points, t, x0, acts, nullptr, workspace, workspace_bytes, side, stream);
}

}  // extern "C"

namespace skgs {
namespace {
int forward_impl(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t, float* x0,
    float* acts, float* out, void* workspace, size_t workspace_bytes, const skgs_adam_range* side, skgs_stream_t stream) {
  Plan p;
  if (make_plan(d, &p)) return 1;
  SKGS_REQUIRE(points && (t || d->t_dim == 0) && acts && workspace, "deform_mlp_forward: NULL argument");
  for (int j = 0; j < d->n_heads; ++j) SKGS_REQUIRE(d->head_out[j], "deform_mlp_forward: head_out[%d] is NULL", j);
  SKGS_REQUIRE(out || d->n_heads, "deform_mlp_forward: no output tensor");
  SKGS_REQUIRE(workspace_bytes >= HDR_BYTES + 2 * p.exch_bytes, "deform_mlp_forward: workspace too small");
  SKGS_REQUIRE((reinterpret_cast<uintptr_t>(acts) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
      "deform_mlp_forward: acts must be 16-byte, the workspace 256-byte aligned");
  FusedArgs a{};
  fill_args(d, p, &a);
  a.points = points, a.t = t, a.x0 = x0, a.acts = acts, a.out = out;
  a.hdr  = reinterpret_cast<unsigned*>(workspace);
  a.exch = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + HDR_BYTES);
  a.census = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a.exch) + p.exch_bytes - CENSUS_BYTES);
  if (bones) {
    if (fill_chain(d, p, bones, &a)) return 1;
    SKGS_REQUIRE(bones->bone_T, "skeleton_forward: bone_T is NULL");
    SKGS_REQUIRE(chain::forward_scratch_floats(bones->M, bones->num_levels) <= (size_t) NW * p.Bp * NC,
        "skeleton_forward: %d tree levels do not fit the scratch", bones->num_levels);
  }
  if (fill_side(side, &a)) return 1;
  const int XW = pad32(p.IN);
  size_t fl = (size_t) p.Bp * (XW + 4) + (size_t) p.Bp * HP + (size_t) NW * p.Bp * NC + (size_t) p.Bp * 4 + (size_t) KL * NC + 4;
  for (int l = 0; l < d->n_layers; ++l) fl += (size_t) NC * ((l ? H : 0) + (d->layer[l].in_x0 ? XW : 0) + 4);
  a.lds_floats = (int) fl;
  const size_t lds = fl * 4;
  SKGS_REQUIRE(lds <= 160 * 1024, "deform_mlp_forward: %zu bytes of LDS needed", lds);
  SKGS_MLP_DISPATCH(fused_mlp_forward_kernel, K_SKELETON_FWD)
}
}  // namespace
}  // namespace skgs

extern "C" {

int skgs_deform_mlp_backward(const skgs_mlp_desc* d, const float* points, const float* t, const float* x0, const float* acts,
    const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  return backward_impl(d, nullptr, points, t, x0, acts, g_out, g_x0, workspace, workspace_bytes, nullptr, stream);
}

int skgs_deform_mlp_backward_adam(const skgs_mlp_desc* d, const float* points, const float* t, const float* x0,
    const float* acts, const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes, const skgs_adam_range* side,
    skgs_stream_t stream) {
  return backward_impl(d, nullptr, points, t, x0, acts, g_out, g_x0, workspace, workspace_bytes, side, stream);
}

int skgs_skeleton_backward(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    const float* x0, const float* acts, float* g_x0, void* workspace, size_t workspace_bytes, const skgs_adam_range* side,
    skgs_stream_t stream) {
  SKGS_REQUIRE(bones, "skeleton_backward: NULL bone chain");
  return backward_impl(d, bones, points, t, x0, acts, nullptr, g_x0, workspace, workspace_bytes, side, stream);
}

}  // extern "C"

namespace skgs {
namespace {
int backward_impl(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    const float* x0, const float* acts, const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes,
    const skgs_adam_range* side, skgs_stream_t stream) {
  Plan p;
  if (make_plan(d, &p)) return 1;
  SKGS_REQUIRE(points && (t || d->t_dim == 0) && acts && workspace, "deform_mlp_backward: NULL argument");
  for (int j = bones ? 1 : 0; j < d->n_heads; ++j)
    SKGS_REQUIRE(d->head_gout[j], "deform_mlp_backward: head_gout[%d] is NULL", j);
  SKGS_REQUIRE(g_out || d->n_heads, "deform_mlp_backward: no incoming gradient");
  SKGS_REQUIRE(workspace_bytes >= HDR_BYTES + 2 * p.exch_bytes, "deform_mlp_backward: workspace too small");
  SKGS_REQUIRE((reinterpret_cast<uintptr_t>(acts) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
      "deform_mlp_backward: acts must be 16-byte, the workspace 256-byte aligned");
  for (int l = 0; l < d->n_layers; ++l) SKGS_REQUIRE(d->layer[l].gW, "deform_mlp_backward: layer %d has no gW", l);
  SKGS_REQUIRE(!g_x0 || p.IN <= d->hidden, "deform_mlp_backward: the input gradient needs hidden >= encoded width (%d < %d)",
      d->hidden, p.IN);
  FusedArgs a{};
  fill_args(d, p, &a);
  a.points = points, a.t = t, a.x0 = const_cast<float*>(x0), a.acts = const_cast<float*>(acts), a.g_out = g_out, a.g_x0 = g_x0;
  a.hdr = reinterpret_cast<unsigned*>(workspace);
  if (bones) {
    if (fill_chain(d, p, bones, &a)) return 1;
    SKGS_REQUIRE(bones->sk_r_raw && bones->chain_A && bones->g_bone_T, "skeleton_backward: sk_r_raw / chain_A / g_bone_T is NULL");
    SKGS_REQUIRE(chain::backward_scratch_floats(bones->M, bones->num_levels) <= (size_t) NW * p.Bp * NC,
        "skeleton_backward: %d tree levels do not fit the scratch", bones->num_levels);
    a.chain.sk_r_raw = bones->sk_r_raw, a.chain.g_bone_T = bones->g_bone_T, a.chain.g_joints = bones->g_joints;
    a.chain.g_global_T = bones->g_global_T, a.chain.g_sk_r_raw = const_cast<float*>(d->head_gout[0]);
  }
  if (fill_side(side, &a)) return 1;
  a.hdr  = reinterpret_cast<unsigned*>(workspace);
  a.exch = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + HDR_BYTES + p.exch_bytes);
  a.census = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a.exch) + p.exch_bytes - CENSUS_BYTES);
  size_t fl = (size_t) p.Bp * HP + (size_t) (d->n_layers + NW) * p.Bp * NC + (size_t) p.Bp * 4 + 4;
  fl += (size_t) 4 * KL;  // the hop loop's per-layer table: gW / gb pointers
  for (int l = 1; l < d->n_layers; ++l) fl += (size_t) NC * (pad32(d->layer[l].out) + 4);
  if (g_x0)
    for (int l = 0; l < d->n_layers; ++l)
      if (d->layer[l].in_x0) fl += (size_t) NC * (pad32(d->layer[l].out) + 4);
  a.lds_floats = (int) fl;
  const size_t lds = fl * 4;
  SKGS_REQUIRE(lds <= 160 * 1024, "deform_mlp_backward: %zu bytes of LDS needed", lds);
  SKGS_MLP_DISPATCH(fused_mlp_backward_kernel, K_SKELETON_BWD)
}
}  // namespace
}  // namespace skgs

extern "C" {

int skgs_deform_mlp_status(const void* workspace, uint32_t* host_words4, skgs_stream_t stream) {
  SKGS_REQUIRE(workspace && host_words4, "deform_mlp_status: NULL argument");
  SKGS_CHECK_HIP(hipMemcpyAsync(host_words4, workspace, 16, hipMemcpyDeviceToHost, (hipStream_t) stream));
  return 0;
}

}  // extern "C"
