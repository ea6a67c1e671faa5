// skgs_common.h -- private layouts and device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include "../../include/skgs.h"

namespace skgs {

constexpr int TILE      = SKGS_TILE;  // 16x16 pixel tiles (reference BLOCK_X/BLOCK_Y)
constexpr int WAVE      = 64;         // CDNA wavefront
constexpr int REC_F4    = 3;          // one Gaussian record = 3 x float4 = 48 B
constexpr int GRAD_ROW  = 16;         // one gradient-accumulator row = 16 floats = one 64-B line

// ---- geom buffer ---------------------------------------------------------------------------------------
// [0,256)   GeomHeader
// [256, ..) GaussRec[P]   (48 B each, 16-B aligned)
//   f[0] x_pix  f[1] y_pix  f[2] conic.a  f[3] conic.b | f[4] conic.c  f[5] opacity  f[6] r  f[7] g |
//   f[8] b      f[9] depth  i[10] radius | clamped_bits << 28   f[11] rcut2 (squared cut-off radius, see preprocess)
struct GeomHeader {
  int32_t num_rendered;
  int32_t overflow;
  int32_t max_tile_count;
  int32_t overflow_events;  // never reset by the library: forwards whose R exceeded the capacity since the caller zeroed it
  int32_t P;
  int32_t big_tiles;        // tiles whose list is longer than the one-wave sort handles (worklist length, per forward)
  int32_t pad[58];
};
static_assert(sizeof(GeomHeader) == 256, "header");

struct GeomView {
  GeomHeader* hdr;
  float4* recs;
};
__host__ __device__ inline GeomView geom_view(void* base) {
  GeomView v;
  v.hdr  = reinterpret_cast<GeomHeader*>(base);
  v.recs = reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + 256);
  return v;
}
inline size_t geom_bytes(int32_t P) { return 256 + (size_t) P * 48 + 256; }

// ---- img buffer ----------------------------------------------------------------------------------------
// n_contrib[H*W] u32 | tile_counts[T] | tile_offsets[T+1] | cursors[T] | tile_begin[T] | tile_end[T] | worklist[T] |
// group_order[ceil(T/8)] (groups of 8 consecutive tiles, longest first: the order the blend kernels walk them in)
// (each 256-B aligned).  tile_begin / tile_end are what every consumer of the lists reads: the compact layout fills
// them from the scan (begin = offsets[t], end = offsets[t+1]), the bucket layout (skgs_raster_inputs::
// tile_bucket_capacity) from the per-tile cursors (begin = t * Lcap).
struct ImgView {
  uint32_t* n_contrib;
  uint32_t* tile_counts;
  uint32_t* tile_offsets;
  uint32_t* cursors;
  uint32_t* tile_begin;
  uint32_t* tile_end;
  uint32_t* worklist;
  uint32_t* group_order;
  int tiles_x, tiles_y, T;
};
constexpr int TILE_GROUP = 8;  // consecutive tiles that travel together through xcd_remap (one XCD, one L2)
__host__ __device__ inline int tile_groups(int T) { return (T + TILE_GROUP - 1) / TILE_GROUP; }
struct TileRanges {
  const uint32_t* begin;
  const uint32_t* end;
  const uint32_t* group_order;  // [ceil(T/8)]: permutation of the tile groups, heaviest first (binning.hip::tile_order_job)
};
// work item v (after xcd_remap) of a blend launch -> (tile, sub): the groups of 8 tiles are walked in `group_order`
template <int SUBS>
__device__ __forceinline__ bool blend_work_item(int v, int T, const uint32_t* __restrict__ group_order, int& tile, int& sub) {
  constexpr int PER_GROUP = TILE_GROUP * SUBS;
  const int g = v / PER_GROUP, j = v % PER_GROUP;
  if (g >= tile_groups(T)) return false;
  tile = (int) group_order[g] * TILE_GROUP + j / SUBS;
  sub  = j % SUBS;
  return tile < T;
}
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
__host__ __device__ inline size_t align256_hd(size_t x) { return (x + 255) & ~size_t(255); }
inline ImgView img_view(void* base, int W, int H) {
  ImgView v;
  v.tiles_x = (W + TILE - 1) / TILE;
  v.tiles_y = (H + TILE - 1) / TILE;
  v.T       = v.tiles_x * v.tiles_y;
  char* p   = reinterpret_cast<char*>(base);
  v.n_contrib = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) W * H * 4);
  v.tile_counts = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) v.T * 4);
  v.tile_offsets = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) (v.T + 1) * 4);
  v.cursors = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) v.T * 4);
  v.tile_begin = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) v.T * 4);
  v.tile_end = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) v.T * 4);
  v.worklist = reinterpret_cast<uint32_t*>(p);
  p += align256((size_t) v.T * 4);
  v.group_order = reinterpret_cast<uint32_t*>(p);
  return v;
}
inline size_t img_bytes(int W, int H) {
  int T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  return align256((size_t) W * H * 4) + align256((size_t) T * 4) + align256((size_t) (T + 1) * 4) +
         4 * align256((size_t) T * 4) + align256((size_t) tile_groups(T) * 4) + 256;
}

// ---- binning buffer ------------------------------------------------------------------------------------
// keys[cap] u64 (depth_bits << 32 | gaussian id; sorted in place per tile) | point_list[cap] u32
struct BinView {
  uint64_t* keys;
  uint32_t* point_list;
  int64_t capacity;
};
inline int64_t bin_capacity(size_t bytes) {
  if (bytes < 512) return 0;
  return (int64_t) ((bytes - 512) / 12);
}
inline size_t bin_bytes(int64_t cap) { return (size_t) cap * 12 + 512; }
inline BinView bin_view(void* base, size_t bytes) {
  BinView v;
  v.capacity   = bin_capacity(bytes);
  char* p      = reinterpret_cast<char*>(base);
  v.keys       = reinterpret_cast<uint64_t*>(p);
  v.point_list = reinterpret_cast<uint32_t*>(p + align256((size_t) v.capacity * 8));
  return v;
}

// ---- XCD-aware block -> work remap -------------------------------------------------------------------
// Blocks are dealt round-robin over the 8 XCDs (block b and b+8 share an XCD / L2).  Work items (the waves of the
// tiles, in raster order) are handed to the XCDs in groups of XCD_GROUP consecutive items: the 4 waves of a tile and
// a few neighbouring tiles (which share Gaussians) hit the same L2, while every XCD still gets an even mix of image
// regions.  One contiguous span per XCD -- the first design -- gave the two XCDs holding the image centre twice the
// work of the ones holding the borders (measured: backward 184 -> 164 us, forward 67 -> 59 us with groups of 8 tiles;
// group sizes 8..128 are within noise).  Pure performance: any placement gives the same results.
#ifndef SKGS_XCD_GROUP
#define SKGS_XCD_GROUP 32
#endif
constexpr int XCD_GROUP = SKGS_XCD_GROUP;  // consecutive work items that stay on one XCD
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int x = b & 7, i = b >> 3;
  return ((i / XCD_GROUP) * 8 + x) * XCD_GROUP + (i % XCD_GROUP);  // may be >= n for the tail: caller must bounds-check
}
// grid size that covers work items 0..n-1 under xcd_remap
inline int xcd_grid(int n) { return ((n + 8 * XCD_GROUP - 1) / (8 * XCD_GROUP)) * (8 * XCD_GROUP); }

// ---- wave-level sum over 64 lanes using DPP (result valid in lane 63) --------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov<0x111, 0xf, 0xf, true>(v);  // row_shr:1
  v += dpp_mov<0x112, 0xf, 0xf, true>(v);  // row_shr:2
  v += dpp_mov<0x114, 0xf, 0xf, true>(v);  // row_shr:4  (lanes 3.. hold sums of 4; after this lane 7,15 hold 8)
  v += dpp_mov<0x118, 0xf, 0xf, true>(v);  // row_shr:8  -> lane 15 of each row = row sum
  v += dpp_mov<0x142, 0xa, 0xf, false>(v); // row_bcast:15 into rows 1,3
  v += dpp_mov<0x143, 0xc, 0xf, false>(v); // row_bcast:31 into rows 2,3 -> lane 63 = total
  return v;
}

// Per-row accumulation into an LDS table with the wave's duplicates merged first.
// Every lane holds N values for table row `row`; LDS float atomics retire about one LANE per 2-3 clocks (deform backward at
// M = 512: 7 M lane-atomics = 37 us of LDS time per CU, tools/pmc_kernel.sh), and lanes that share a row serialise on top.
// With the Gaussians in spatial order (sk_gs_amd/densify.py::sort_spatially) the 64 lanes of a wave name only a handful of
// distinct rows: for each of the first WAVE_GROUPS_MAX distinct rows the group's N sums are formed by masked DPP wave
// reductions and added by ONE ds_add_f32 with N active lanes (distinct addresses); lanes of any further row fall back to
// their own atomics.  ALL 64 lanes must be active (`valid` masks the ones without work).  Sum order inside a group is the DPP
// tree's: deterministic for a given lane assignment.
constexpr int WAVE_GROUPS_MAX = 6;
template <int N>
__device__ __forceinline__ void wave_group_add(float* __restrict__ table, int stride, int row, const float (&v)[N], bool valid) {
  const int lane = threadIdx.x & 63;
  uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll 1
  for (int it = 0; todo != 0 && it < WAVE_GROUPS_MAX; ++it) {
    const int leader    = __builtin_ctzll(todo);
    const int r         = __builtin_amdgcn_readlane(row, leader);
    const uint64_t same = __builtin_amdgcn_ballot_w64(valid && row == r) & todo;
    const bool mine     = (same >> lane) & 1;
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const float x = wave_sum_to_lane63(mine ? v[c] : 0.f);
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
      tot = lane == c ? t : tot;
    }
    if (lane < N) atomicAdd(table + (size_t) r * stride + lane, tot);
    todo &= ~same;
  }
  if ((todo >> lane) & 1) {
#pragma unroll
    for (int c = 0; c < N; ++c) atomicAdd(table + (size_t) row * stride + c, v[c]);
  }
}

// Three independent wave sums at once, as hand-placed v_add_f32_dpp (one instruction per step and value: hipcc's
// DPP-combine leaves most update_dpp + add pairs un-fused, doubling the instruction count of the reduction).
// A VALU write -> DPP read of the same VGPR needs 2 wait states on gfx9-family parts: inside the statement the
// three chains are interleaved, so each value's next step is exactly two instructions after its previous one; the
// leading s_nop covers values produced by compiler code just before the statement.  Totals land in lane 63.
#define SKGS_DPP3_STEP(CTRL)                                   \
  "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                      \
  "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                      \
  "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"
__device__ __forceinline__ void wave_sum3_to_lane63(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
      SKGS_DPP3_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      SKGS_DPP3_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      SKGS_DPP3_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      SKGS_DPP3_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      SKGS_DPP3_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
      SKGS_DPP3_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 0"
      : "+v"(a), "+v"(b), "+v"(c));
}

// Nine wave sums at once by "transposing" the reduction: every halving step packs two values into one register, so the work
// shrinks with the lane span instead of being 9 x 6 DPP adds.  The steps that stay inside a row of 16 lanes are DPP adds
// under a BANK mask:
//   v_add_f32_dpp r, a, a row_mirror      bank_mask:0x3   lanes 0-7  of every row: a[l] + a[15 - l]
//   v_add_f32_dpp r, b, b row_mirror      bank_mask:0xc   lanes 8-15 of every row: b[l] + b[15 - l]
// = two values in one register for two instructions of 4.2 issue clocks each (tools/micro/valu_issue_rate.hip), where a
// v_permlane swap + add is 8.2 + 2.6; row_half_mirror with banks 0x5 / 0xa does the next halving.  Eight values are then in
// two registers (8 + 4 DPP adds), one v_permlane32_swap + add joins those, the ninth value (two full-row mirror adds) is
// joined by ONE v_permlane16_swap + add, two quad_perm adds finish the rows and a row_bcast:31 add the ninth:
// 17 DPP adds, 2 swaps, 2 adds.  (Round 1-2 version: both wide halvings as v_permlane32/16_swap + add, the ninth value as a
// separate 6-step DPP chain: 11 DPP adds, 6 swaps, 8 adds, 3 selects -- blend backward 121 -> 112 us at config #1.)
// A VALU write followed by a DPP / permlane read of the same VGPR needs 2 wait states and the compiler cannot see inside
// the statement: the independent chains are interleaved and s_nop covers the rest.
// All 64 lanes must be active.  On return `v0` holds every total: lane 4 m + 32 h (m = 0..3, h = 0..1) that of value
// banked_holder_value(lane), lanes 48-63 that of v8; v1..v8 are clobbered.
__device__ __forceinline__ void wave_sum9_banked(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6,
                                                 float& v7, float& v8) {
#define SKGS_MIRROR(dst, src, banks) "v_add_f32_dpp " dst ", " src ", " src " row_mirror row_mask:0xf bank_mask:" banks "\n\t"
#define SKGS_HALF(dst, src, banks) "v_add_f32_dpp " dst ", " src ", " src " row_half_mirror row_mask:0xf bank_mask:" banks "\n\t"
  asm volatile(
      "s_nop 1\n\t"
      SKGS_MIRROR("%0", "%0", "0x3") SKGS_MIRROR("%2", "%2", "0x3") SKGS_MIRROR("%4", "%4", "0x3") SKGS_MIRROR("%6", "%6", "0x3")
      SKGS_MIRROR("%0", "%1", "0xc") SKGS_MIRROR("%2", "%3", "0xc") SKGS_MIRROR("%4", "%5", "0xc") SKGS_MIRROR("%6", "%7", "0xc")
      SKGS_MIRROR("%8", "%8", "0xf")
      SKGS_HALF("%0", "%0", "0x5") SKGS_HALF("%4", "%4", "0x5")
      SKGS_HALF("%8", "%8", "0xf")
      SKGS_HALF("%0", "%2", "0xa") SKGS_HALF("%4", "%6", "0xa")
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %4\n\t"
      "v_add_f32 %0, %0, %4\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %8\n\t"
      "v_add_f32 %0, %0, %8\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0x8 bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8));
#undef SKGS_MIRROR
#undef SKGS_HALF
}
// lanes of `v0` that hold a total after wave_sum9_banked, and which one
__device__ __forceinline__ bool banked_holder(int lane) { return (lane & 0x13) == 0 || lane == 48; }
__device__ __forceinline__ int banked_holder_value(int lane) {
  return lane == 48 ? 8 : ((lane >> 5) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 3) & 1);
}

// Tile rectangle of a splat (reference getRect, gaussian_render.h:42-47). Used by the preprocess AND the scatter
// kernel: both must produce the identical rectangle.  No multiply feeds an add here, so FMA contraction settings
// of the including file cannot change the result.
__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, int* mn, int* mx) {
  mn[0] = min(gx, max(0, (int) ((px - r) / TILE)));
  mn[1] = min(gy, max(0, (int) ((py - r) / TILE)));
  mx[0] = min(gx, max(0, (int) ((px + r + TILE - 1) / TILE)));
  mx[1] = min(gy, max(0, (int) ((py + r + TILE - 1) / TILE)));
}

// Streaming accesses.  A plain store leaves its line dirty in the XCD's L2 until it is evicted or the launch's closing write-back
// flushes it, and a plain load allocates its line there; for arrays that are touched once per step and not again before tens of MB
// have passed, the non-temporal hint (`global_load / global_store ... nt`) keeps them out of the way.  Measured on one box with
// alternating builds (SKGS_NT_MASK, one bit per site; 10-12 default bench runs each; ms per step, config #1):
//   no site 0.3511 | the optimizer's stores 0.3496 | + the dense logit-gradient rows 0.3481 | + the optimizer's loads 0.3474
// and neutral or worse: the SH gradient rows, the loss maps (read back by the very next launch: image_loss_backward 19.2 -> 21.7 us),
// dL/dimage (the loss backward -0.5 us, the blend backward +1), the SH rows' and the gradient rows' loads.  Default 0x61 = the three
// sites that paid; make CXXFLAGS="... -DSKGS_NT_MASK=0" builds every access plain.
#ifndef SKGS_NT_MASK
#define SKGS_NT_MASK 0x61
#endif
enum StreamSite { NT_ADAM = 0, NT_PRE_BWD = 1, NT_LOSS_FWD = 2, NT_LOSS_BWD = 3, NT_PRE_FWD = 4, NT_DEFORM_BWD = 5, NT_ADAM_LOAD = 6,
  NT_SH_LOAD_FWD = 7, NT_SH_LOAD_BWD = 8, NT_GRADROW_LOAD = 9 };
// SKGS_WT_MASK: sites whose stores are WRITE-THROUGH (`sc1`: the line leaves the XCD's L2 at once and is not kept) -- what shortens a
// launch's closing write-back of the lines it dirtied (a write-heavy launch has 3-6 us at its end with no wave on the chip:
// SQ_BUSY_CYCLES against GRBM_GUI_ACTIVE, less the ~20 k clocks the counter collection adds to every dispatch).
// Measured like the mask above (8 alternating runs each, ms per step): the optimizer's stores nt 0.3460 -> sc1 0.3429; the SH gradient
// rows sc1 0.3476 (worse: the optimizer piece of the NEXT launch reads them, out of L2 while they are there).  sc1 only pays on 16-byte
// stores (a dword sc1 store is a fabric write of its own, MI355X_MICROARCH.md).  Default 0x1 = the optimizer's three output arrays.
#ifndef SKGS_WT_MASK
#define SKGS_WT_MASK 0x1
#endif
template <int SITE>
__device__ __forceinline__ void stream_store(float* p, float v) {
  if constexpr (SITE < 31 && ((SKGS_WT_MASK >> SITE) & 1)) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (SITE < 31 && ((SKGS_NT_MASK >> SITE) & 1)) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int SITE>
__device__ __forceinline__ float4 stream_load4(const float* p) {
  if constexpr (SITE < 31 && ((SKGS_NT_MASK >> SITE) & 1)) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <int SITE>
__device__ __forceinline__ void stream_store4(float* p, float4 v) {
  if constexpr (SITE < 31 && ((SKGS_WT_MASK >> SITE) & 1)) {
    typedef float f4w __attribute__((ext_vector_type(4)));
    const f4w q = {v.x, v.y, v.z, v.w};
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx941__) || !defined(__HIP_DEVICE_COMPILE__)
    // (the `sc1` modifier exists on gfx94x / gfx950 only -- this library is written for gfx950, csrc/Makefile ARCH; the statement is a
    // store the compiler's waitcnt bookkeeping does not see: the `memory` clobber keeps it ordered against the surrounding accesses and
    // vmcnt retires in order, so the kernel's closing s_endpgm wait covers it)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
#else
    __builtin_nontemporal_store(q, reinterpret_cast<f4w*>(p));     // another target: no write-through form; the streaming hint instead
#endif
  } else if constexpr (SITE < 31 && ((SKGS_NT_MASK >> SITE) & 1)) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

}  // namespace skgs

// host-side launch declarations (one per TU)
namespace skgs {
int set_error(const char* fmt, ...);
#define SKGS_CHECK_HIP(expr)                                                                 \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return skgs::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)
#define SKGS_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return skgs::set_error(__VA_ARGS__); \
  } while (0)

// ---- optional per-kernel HIP-event timing (api.hip). Off by default; bench.py turns it on for the kernels it
// reports.  Events are recorded on the stream the kernel is launched on.
enum KernelId {
  K_PREPROCESS_FWD = 0, K_SCAN, K_SCATTER, K_SORT, K_RENDER_FWD, K_RENDER_BWD, K_PREPROCESS_BWD, K_DEFORM_FWD,
  K_DEFORM_BWD, K_KNN, K_LOSS_FWD, K_LOSS_BWD, K_SKELETON_FWD, K_SKELETON_BWD, K_ADAM, K_SP_NET_FWD, K_SP_NET_BWD, K_SP_KNN,
  K_SP_KNN_BWD, K_COUNT
};
void prof_begin(int kid, hipStream_t s);
void prof_end(int kid, hipStream_t s);
struct ProfScope {
  int kid;
  hipStream_t s;
  ProfScope(int k, hipStream_t st) : kid(k), s(st) { prof_begin(kid, s); }
  ~ProfScope() { prof_end(kid, s); }
};

// ---- inverse neighbour lists of the superpoint stage (sp_knn.hip files them, sp_backward.hip walks them) ------------------
// [0,256) header {cap, overflow flag of the last forward, overflow EVENTS}, counts[M] (256-B aligned), lists[M][cap] of pair ids
// (n << 4 | k).  Capacity per superpoint: 16 x the mean list (P K / M), at least 4096, at most P: a list that outgrows it raises the
// overflow flag (cleared by every forward's preparation) and, once per forward, bumps the event counter in word 2, which the
// library never clears (as GeomHeader::overflow_events): a training loop that looks every N steps still sees a dropped pair.
struct SpPairsView {
  uint32_t* header;  // [0] cap, [1] overflow (this forward), [2] forwards that overflowed since the caller zeroed the buffer
  uint32_t* counts;
  uint32_t* lists;
  float* table;      // [M][12]: the superpoints as the search's LDS rows (xyz | 8 hyper | id), in scan order
  int cap;
};
// The superpoint table of the search packed once per step -- rows [xyz | hyper | id] in scan order, 12 floats each -- and the
// inverse lists' header and counters cleared: the job of sp_knn.hip::sp_prepare_kernel, or of extra workgroups of the network's
// weight-transposition launch in front of it (sp_mlp.hip: both are per-step preparations of parameters, neither needs the other)
struct SpPrepareJob {
  int M, F;
  const float* sp_points;
  const float* sp_feature;
  const int32_t* sp_order;
  uint32_t* header_and_counts;
  int n_clear;
  float* table;
};
#if defined(__HIPCC__)
__device__ __forceinline__ void sp_prepare_element(const SpPrepareJob& j, int i) {
  constexpr int ROW = 12;
  if (i < j.n_clear && i != 2) j.header_and_counts[i] = 0u;  // (word 2: the sticky overflow-event counter)
  if (i >= j.M * ROW) return;
  const int r = i / ROW, c = i - r * ROW;
  const int id = j.sp_order ? j.sp_order[r] : r;
  float v = 0.f;
  if (c < 3)
    v = j.sp_points[3 * id + c];
  else if (c < 3 + j.F)
    v = j.sp_feature[(size_t) id * j.F + c - 3];
  else if (c == ROW - 1)
    v = __builtin_bit_cast(float, id);
  j.table[i] = v;
}
#endif
inline size_t sp_pairs_capacity(int P, int M, int K) {
  const size_t mean = ((size_t) P * K + M - 1) / M;
  return std::min<size_t>((size_t) std::max(P, 1), std::max<size_t>(4096, 16 * mean));
}
inline SpPairsView sp_pairs_view(void* base, int P, int M, int K) {
  SpPairsView v;
  v.header = reinterpret_cast<uint32_t*>(base);
  v.counts = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + 256);
  v.lists  = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + 256 + align256((size_t) M * 4));
  v.cap    = (int) sp_pairs_capacity(P, M, K);
  v.table  = reinterpret_cast<float*>(v.lists + (size_t) M * v.cap);
  return v;
}
size_t sp_pairs_bytes(int P, int M, int K);  // sp_backward.hip

// api.hip: p[0..n_words) = v  (kernel, not hipMemsetAsync: keeps captured graphs to kernel nodes only)
int fill_u32(void* p, uint32_t v, size_t n_words, hipStream_t s);

// preprocess.hip
int launch_preprocess_forward(const skgs_raster_inputs& in, GeomView g, ImgView im, int32_t* radii, hipStream_t s);
int launch_preprocess_backward(const skgs_raster_inputs& in, GeomView g, const int32_t* radii,
    const skgs_raster_grads& gr, hipStream_t s);
int launch_sh_grad_from_factors(int P, int n_views, int D, int M, const float* factors, float* dL_dsh, float* dL_dsh_rest,
    hipStream_t s);
int launch_mark_visible(int P, const float* means, const float* view, int colmap, uint8_t* present, hipStream_t s);
// binning.hip
int launch_scan_tiles(GeomView g, ImgView im, int64_t P, hipStream_t s);  // count tiles (16 lanes / Gaussian) + scan
int launch_scatter_sort(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, hipStream_t s);
// render.hip
bool gradacc_rows_hold_moments();
int launch_render_forward(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, float* out_color,
    float* out_opacity, float* out_extra, hipStream_t s);
int launch_render_census(int W, int H, GeomView g, ImgView im, BinView b, float* out_color, float* out_opacity,
    uint32_t* census, hipStream_t s);
int launch_render_backward(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, const float* out_opacity,
    const float* dL_dcolor, const float* dL_dopacity, const float* dL_dextra, float* gradacc, hipStream_t s);
int launch_extra_forward(int W, int H, int P, int E, const float* extra, GeomView g, ImgView im, BinView b,
    float* pixel_extra, hipStream_t s);
int launch_extra_backward(int W, int H, int P, int E, const float* extra, const float* out_opacity,
    const float* grad_pixel_extra, GeomView g, ImgView im, BinView b, float* grad_means2D, float* grad_conic,
    float* grad_opacity, float* dL_dextra, hipStream_t s);
int launch_topk(int topk, int W, int H, GeomView g, ImgView im, BinView b, int32_t* top_idx, float* top_w,
    hipStream_t s);
// deform.hip
int launch_deform_forward(const skgs_deform_inputs& in, float* means, float* scales, float* rotations, float* opacity,
    float* d_xyz, float* d_rot, float* d_scale, hipStream_t s);
size_t deform_backward_workspace_bytes(int P, int M);
int launch_deform_backward(const skgs_deform_inputs& in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, void* workspace,
    hipStream_t s, float* g_sp_W = nullptr, float* g_logits = nullptr);
int launch_knn_bones(int P, int M, int K, int dim, const float* points, const float* joints, float* out_dist,
    int64_t* out_idx, hipStream_t s);
int launch_knn_dist_weights_forward(int P, int M, int K, int dim, const float* points, const float* joints, const float* radius,
    const float* kweight, float temperature, int activate, int64_t* out_idx, float* out_weights, float* out_dist, hipStream_t s);
size_t knn_dist_weights_workspace_bytes(int P, int M, int dim);
int launch_knn_dist_weights_backward(int P, int M, int K, int dim, const float* points, const float* joints, const float* radius,
    const float* kweight, float temperature, int activate, int accumulate_joints, const float* weights, const int64_t* indices,
    const float* nn_dist, const float* g_weights, float* g_points, float* g_joints, float* g_radius, float* g_kweight,
    void* workspace, size_t workspace_bytes, hipStream_t s);
int launch_knn_lbs_weights(int P, int M, int K, const float* points, const float* joints, const float* sp_W, int64_t* out_idx,
    float* out_weights, hipStream_t s);
int sp_skinning_check(const skgs_sp_skinning_job& j);                       // sp_backward.hip
int launch_sp_skinning_rest(const skgs_sp_skinning_job& j, hipStream_t s);  // bones + finalize
int launch_deform_backward_finalize(const skgs_deform_inputs& in, void* workspace, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, hipStream_t s);
int deform_backward_job_max_bones();
int deform_backward_job_max_k();
int launch_knn_deform_forward(int P, int M, int K, const float* points, const float* joints, const float* sp_W,
    const float* bone_T, const float* bone_drot, const float* bone_dscale, const float* xyz, const float* log_scale,
    const float* rot, const float* opacity_logit, int64_t* out_idx, float* out_weights, float* means, float* scales,
    float* rotations, float* opacity, const int32_t* live_count, hipStream_t s);
int launch_lbs_weights_backward_compact(int P, int K, const float* weights, const float* g_weights, float* g_logits,
    hipStream_t s);
int launch_lbs_logits_scatter(int P, int M, int K, const int64_t* indices, const float* g_logits, float* g_sp_W,
    hipStream_t s);
int launch_lbs_weights_forward(int P, int M, int K, const float* sp_W, const int64_t* indices, float* weights, hipStream_t s);
int launch_lbs_weights_backward(int P, int M, int K, const float* weights, const int64_t* indices, const float* g_weights,
    float* g_sp_W, hipStream_t s);
}  // namespace skgs
