// binning.hip -- builds the per-tile, depth-sorted Gaussian lists (gfx950).
//
// Reference pipeline (gaussian_rasterizer_forward.cu:45-94,203-241): InclusiveSum over Gaussians -> D2H copy of R ->
// duplicateWithKeys (64-bit key = tile<<32 | depth bits) -> ONE global 64-bit radix sort of all R pairs ->
// identifyTileRanges.  Resulting order inside a tile: ascending raw depth bits, ties by ascending Gaussian id
// (CUB's radix sort is stable and instances are emitted in Gaussian order).
//
// Here the tile is not sorted, it is *addressed*: tile counts were accumulated by the preprocess kernel, an
// exclusive scan over the T tiles gives every tile its final [start,end) range directly (= identifyTileRanges),
// instances are scattered into their tile's range, and each tile then sorts its own short list in LDS on the
// 64-bit key depth_bits<<32 | id  -- the same total order, with 1 pass over the R instances instead of ~6 radix
// passes, no host round trip and nothing whose launch shape depends on R (hipGraph-capturable).
#include <algorithm>
#include <cstdlib>

#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int SCAN_THREADS = 1024;

// Single-workgroup exclusive scan of tile_counts[T] -> tile_offsets[T+1]; zeroes cursors; publishes R.
__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(int T, const uint32_t* __restrict__ counts,
    uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors, uint32_t* __restrict__ tile_begin,
    uint32_t* __restrict__ tile_end, GeomHeader* hdr) {
  __shared__ uint32_t wave_tot[SCAN_THREADS / WAVE];
  __shared__ uint32_t wave_max[SCAN_THREADS / WAVE];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  uint32_t mx = 0;
  __syncthreads();
  for (int base = 0; base < T; base += SCAN_THREADS) {
    const int i      = base + tid;
    const uint32_t c = i < T ? counts[i] : 0u;
    mx               = max(mx, c);
    uint32_t v       = c;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(v, d);
      if (lane >= d) v += o;
    }
    if (lane == 63) wave_tot[wid] = v;
    __syncthreads();
    uint32_t wprefix = 0;
    for (int w = 0; w < wid; ++w) wprefix += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < T) {
      offsets[i] = carry + wprefix + v - c;
      cursors[i] = 0;
      tile_begin[i] = carry + wprefix + v - c, tile_end[i] = carry + wprefix + v;
    }
    __syncthreads();
    if (tid == SCAN_THREADS - 1) carry_s = carry + wprefix + v;
    __syncthreads();
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t) __shfl_xor((int) mx, d));
  if (lane == 0) wave_max[wid] = mx;
  __syncthreads();
  if (tid == 0) {
    uint32_t m = 0;
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) m = max(m, wave_max[w]);
    offsets[T]          = carry_s;
    hdr->num_rendered   = (int32_t) carry_s;
    hdr->max_tile_count = (int32_t) m;
    hdr->overflow       = 0;
    hdr->big_tiles      = 0;
  }
}

// Tile counting and scatter: LPG lanes per Gaussian.  A Gaussian touches ~5 tiles on average but the loops of a
// one-lane-per-Gaussian kernel are serial chains of (returning) atomics on a grid of only ~1.5 waves per SIMD, i.e.
// pure latency (measured: 84 us for 5e5 atomics).  With 16 lanes per Gaussian every (Gaussian, tile) pair is its own
// lane, all atomics of a wave are in flight together and there are 16x more waves to hide their latency.
constexpr int LPG = 16;

// Adds 1 to tile_counts[t] for every tile the splat's rectangle covers.
__global__ void __launch_bounds__(256) count_tiles_kernel(int P, int gx, int gy, const float4* __restrict__ recs,
    uint32_t* __restrict__ tile_counts) {
  const int64_t tid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  const int idx     = (int) (tid / LPG);
  if (idx >= P) return;
  const float4 r2  = recs[3 * idx + 2];
  const int radius = __float_as_int(r2.z) & 0x0fffffff;
  if (radius <= 0) return;
  const float4 r0 = recs[3 * idx];
  int mn[2], mx[2];
  tile_rect(r0.x, r0.y, radius, gx, gy, mn, mx);
  const int w = mx[0] - mn[0], n = w * (mx[1] - mn[1]);
  for (int k = (int) (tid % LPG); k < n; k += LPG) atomicAdd(&tile_counts[(mn[1] + k / w) * gx + mn[0] + k % w], 1u);
}

// Writes (depth_bits<<32 | id) into every touched tile's range.
__global__ void __launch_bounds__(256) scatter_kernel(int P, int gx, int gy, const float4* __restrict__ recs,
    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors, uint64_t* __restrict__ keys, int64_t capacity,
    GeomHeader* hdr, int bucket) {
  const int64_t tid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (!bucket && tid == 0 && (int64_t) hdr->num_rendered > capacity) hdr->overflow = 1, hdr->overflow_events += 1;
  const int idx = (int) (tid / LPG);
  if (idx >= P) return;
  const float4 r2  = recs[3 * idx + 2];
  const int radius = __float_as_int(r2.z) & 0x0fffffff;
  if (radius <= 0) return;
  const float4 r0 = recs[3 * idx];
  int mn[2], mx[2];
  tile_rect(r0.x, r0.y, radius, gx, gy, mn, mx);
  const uint64_t key = ((uint64_t) __float_as_uint(r2.y) << 32) | (uint32_t) idx;
  const int w = mx[0] - mn[0], n = w * (mx[1] - mn[1]);
  for (int k = (int) (tid % LPG); k < n; k += LPG) {
    const int t        = (mn[1] + k / w) * gx + mn[0] + k % w;
    const uint32_t slot = atomicAdd(&cursors[t], 1u);
    if (bucket) {
      if (slot < (uint32_t) bucket) keys[(size_t) t * bucket + slot] = key;
    } else if ((int64_t) offsets[t] + slot < capacity) {
      keys[offsets[t] + slot] = key;
    }
  }
}

// LDS-privatised variants (T <= BIN_LDS_TILES).  Device-scope atomics execute at the memory side on MI355X (the
// per-XCD L2s are not coherent): 5e5 scattered single-dword atomics cost ~45 us whatever the launch shape.  Here a few
// large workgroups histogram their share of the (Gaussian, tile) pairs in LDS (ds_add is ~1 lane/clk) and touch global
// memory once per non-empty bin: ~BIN_GROUPS x T global atomics instead of R.
constexpr int BIN_LDS_TILES = 8192;
constexpr int BIN_THREADS   = 1024;
constexpr int BIN_GROUPS    = 512;  // two per CU (measured: 48 -> 82 us, 256 -> 42 us for count + scatter; round 5, 8 alternating bench
                                    // runs each: 256 0.3470 ms per step, 512 0.3461, 768 / 1024 the same); SKGS_BIN_GROUPS overrides
static int bin_groups() {
  static int v = [] {
    const char* e = getenv("SKGS_BIN_GROUPS");
    const int n   = e ? atoi(e) : 0;
    return n > 0 ? n : BIN_GROUPS;
  }();
  return v;
}

__device__ __forceinline__ bool splat_rect(const float4* __restrict__ recs, int idx, int gx, int gy, int* mn, int& w, int& n,
    uint32_t& depth_bits) {
  const float4 r2  = recs[3 * idx + 2];
  const int radius = __float_as_int(r2.z) & 0x0fffffff;
  if (radius <= 0) return false;
  const float4 r0 = recs[3 * idx];
  int mx[2];
  tile_rect(r0.x, r0.y, radius, gx, gy, mn, mx);
  w          = mx[0] - mn[0];
  n          = w * (mx[1] - mn[1]);
  depth_bits = __float_as_uint(r2.y);
  return n > 0;
}

__global__ void __launch_bounds__(BIN_THREADS) count_tiles_lds_kernel(int P, int gx, int gy, int T,
    const float4* __restrict__ recs, uint32_t* __restrict__ tile_counts) {
  extern __shared__ uint32_t s_cnt[];  // [T]
  for (int i = threadIdx.x; i < T; i += BIN_THREADS) s_cnt[i] = 0;
  __syncthreads();
  const int64_t lanes = (int64_t) P * LPG;
  for (int64_t tid = (int64_t) blockIdx.x * BIN_THREADS + threadIdx.x; tid < lanes; tid += (int64_t) gridDim.x * BIN_THREADS) {
    int mn[2], w, n;
    uint32_t db;
    if (!splat_rect(recs, (int) (tid / LPG), gx, gy, mn, w, n, db)) continue;
    for (int k = (int) (tid % LPG); k < n; k += LPG) atomicAdd(&s_cnt[(mn[1] + k / w) * gx + mn[0] + k % w], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += BIN_THREADS) {
    const uint32_t c = s_cnt[i];
    if (c) atomicAdd(&tile_counts[i], c);
  }
}

template <int LPG>  // lanes per Gaussian of the two tile walks
__global__ void __launch_bounds__(BIN_THREADS) scatter_lds_kernel(int P, int gx, int gy, int T, const float4* __restrict__ recs,
    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors, uint64_t* __restrict__ keys, int64_t capacity,
    GeomHeader* hdr, int bucket /* 0: compact lists at offsets[]; > 0: tile t owns slots [t * bucket, (t + 1) * bucket) */) {
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_cnt  = s_mem;      // [T] entries of this workgroup per tile, then the running local rank
  uint32_t* s_base = s_mem + T;  // [T] offsets[t] + slots reserved for this workgroup
  if (!bucket && blockIdx.x == 0 && threadIdx.x == 0 && (int64_t) hdr->num_rendered > capacity)
    hdr->overflow = 1, hdr->overflow_events += 1;
  for (int i = threadIdx.x; i < T; i += BIN_THREADS) s_cnt[i] = 0;
  __syncthreads();
  const int64_t lanes  = (int64_t) P * LPG;
  const int64_t stride = (int64_t) gridDim.x * BIN_THREADS;
  for (int64_t tid = (int64_t) blockIdx.x * BIN_THREADS + threadIdx.x; tid < lanes; tid += stride) {
    int mn[2], w, n;
    uint32_t db;
    if (!splat_rect(recs, (int) (tid / LPG), gx, gy, mn, w, n, db)) continue;
    for (int k = (int) (tid % LPG); k < n; k += LPG) atomicAdd(&s_cnt[(mn[1] + k / w) * gx + mn[0] + k % w], 1u);
  }
  __syncthreads();
  // slot reservation: one returning global atomic per touched tile.  Four tiles per lane and round, so that the
  // (memory-side, ~2 us) round trips of a lane are in flight together instead of one after the other
  for (int i0 = threadIdx.x; i0 < T; i0 += 4 * BIN_THREADS) {
    uint32_t c[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * BIN_THREADS;
      c[u] = i < T ? s_cnt[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = c[u] ? atomicAdd(&cursors[i0 + u * BIN_THREADS], c[u]) : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * BIN_THREADS;
      if (i < T) {
        s_base[i] = c[u] ? (bucket ? 0u : offsets[i]) + r[u] : 0u;  // bucket: index inside the tile
        s_cnt[i]  = 0;
      }
    }
  }
  __syncthreads();
  for (int64_t tid = (int64_t) blockIdx.x * BIN_THREADS + threadIdx.x; tid < lanes; tid += stride) {
    int mn[2], w, n;
    uint32_t db;
    const int idx = (int) (tid / LPG);
    if (!splat_rect(recs, idx, gx, gy, mn, w, n, db)) continue;
    const uint64_t key = ((uint64_t) db << 32) | (uint32_t) idx;
    for (int k = (int) (tid % LPG); k < n; k += LPG) {
      const int t        = (mn[1] + k / w) * gx + mn[0] + k % w;
      const uint32_t pos = s_base[t] + atomicAdd(&s_cnt[t], 1u);
      if (bucket) {
        if (pos < (uint32_t) bucket) keys[(size_t) t * bucket + pos] = key;  // entries beyond the bucket are dropped (flagged)
      } else if ((int64_t) pos < capacity) {
        keys[pos] = key;
      }
    }
  }
}

// Per-tile sort. Ascending-only bitonic network with virtual +inf padding (works for any length).
constexpr int SORT_THREADS = 256;
constexpr int SORT_LDS_MAX = 8192;  // 64 KB of u64

// `first_size`: the blocks of first_size / 2 keys are already sorted ascending (2 = nothing is)
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_any(KeyPtr k, int L, int n /*pow2 >= L*/, int tid, int first_size = 2) {
  for (int size = first_size; size <= n; size <<= 1) {
    // mirror step: a in lower half of each block, partner = block end - offset
    const int half = size >> 1;
    for (int t = tid; t < (n >> 1); t += SORT_THREADS) {
      const int blk = t / half, r = t - blk * half;
      const int a = blk * size + r, b = blk * size + size - 1 - r;
      if (b < L) {
        const uint64_t ka = k[a], kb = k[b];
        if (ka > kb) k[a] = kb, k[b] = ka;
      }
    }
    __syncthreads();
    for (int j = half >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n >> 1); t += SORT_THREADS) {
        const int a = 2 * j * (t / j) + (t % j), b = a + j;
        if (b < L) {
          const uint64_t ka = k[a], kb = k[b];
          if (ka > kb) k[a] = kb, k[b] = ka;
        }
      }
      __syncthreads();
    }
  }
}

// Lists of up to 64 * E keys (E <= 8: 512 keys, nearly every tile): ONE wave sorts the list entirely in
// registers.  Blocked layout -- lane l holds keys l*E .. l*E+E-1 -- so a bitonic compare-exchange at distance j < E is
// a swap between two registers of the same lane and only distances j >= E need one cross-lane read (lane ^ (j/E),
// ds_bpermute): 21 of the 36 steps for 256 keys.  No LDS array, no barriers.  The LDS network above needed a
// workgroup barrier per step and 32 KB of LDS per tile.  Padding keys are ~0 (above every real key: the low word is a
// Gaussian id < 2^31).
constexpr int WSORT_WAVE_MAX = 512;   // longest list one wave sorts on its own (E = 8)

// Value of lane ^ J for J in {1, 2, 4, 8, 16, 32} without the LDS crossbar (ds_bpermute: ~100 cycles of latency per step,
// and the sort is a chain of dependent steps at ~2.5 waves per SIMD): DPP quad_perm (1, 2), two bank-masked row shifts
// (4), row_ror:8 (8: +8 and -8 coincide in a row of 16), and the gfx950 v_permlane16/32_swap on two copies (16, 32).
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
  const int x = (int) v;
  if constexpr (J == 1) {
    return (uint32_t) __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
  } else if constexpr (J == 2) {
    return (uint32_t) __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
  } else if constexpr (J == 4) {
    int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> lanes 0-3, 8-11 of a row read l+4
    t     = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);  // row_shr:4 -> lanes 4-7, 12-15 read l-4
    return (uint32_t) t;
  } else if constexpr (J == 8) {
    return (uint32_t) __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);  // row_ror:8
  } else if constexpr (J == 16) {
    uint32_t a = v, b = v;
    // odd rows of a <-> even rows of b:  a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return (lane & 16) ? a : b;
  } else {
    static_assert(J == 32, "lane_xor: J must be a power of two <= 32");
    uint32_t a = v, b = v;
    // upper half of a <-> lower half of b:  a = [lo lo], b = [hi hi]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return (lane & 32) ? a : b;
  }
}

// jl is a constant once the network's loops are unrolled: the switch folds away
__device__ __forceinline__ uint32_t lane_xor_any(uint32_t v, int jl, int lane) {
  switch (jl) {
    case 1: return lane_xor<1>(v, lane);
    case 2: return lane_xor<2>(v, lane);
    case 4: return lane_xor<4>(v, lane);
    case 8: return lane_xor<8>(v, lane);
    case 16: return lane_xor<16>(v, lane);
    default: return lane_xor<32>(v, lane);
  }
}

// WAVES = 1: the one-wave network.  WAVES = 4: the same network over the 256 lanes of a workgroup -- "virtual lane"
// vlane = wave * 64 + lane holds keys vlane*E .. vlane*E+E-1; the only steps that leave the wave are the ones at distance
// >= 64*E (3 of the 55 steps for 1024 keys), done as an exchange through LDS (sx: WAVES*64*E keys) with two barriers.
// A 1024-key list costs a wave 4 keys per step instead of 16.
template <int E, int WAVES>
__device__ __forceinline__ void bitonic_blocked(uint64_t (&k)[E], int lane, int wave, uint64_t* __restrict__ sx) {
  constexpr int N = 64 * E * WAVES;
  const int vlane = wave * 64 + lane;
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int j = size >> 1; j > 0; j >>= 1) {
      if (j < E) {  // partner in another register of the same lane (index bits below E)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if ((e & j) == 0) {
            // ascending iff bit `size` of the global index vlane*E + e is clear
            const bool asc   = size < E ? ((e & size) == 0) : (((vlane * E) & size) == 0);
            const uint64_t a = k[e], b = k[e | j];
            const bool sw = (a > b) == asc;  // one 64-bit compare; equal keys (padding) may swap freely
            k[e]          = sw ? b : a;
            k[e | j]      = sw ? a : b;
          }
        }
      } else if (j < 64 * E) {  // partner in lane ^ (j / E), same register
        const int jl     = j / E;
        const bool upper = (lane & jl) != 0;
        const bool asc   = ((vlane * E) & size) == 0;  // size > j >= E: a lane bit
        const bool keep_min = asc != upper;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint64_t a  = k[e];
          const uint32_t lo = lane_xor_any((uint32_t) a, jl, lane), hi = lane_xor_any((uint32_t) (a >> 32), jl, lane);
          const uint64_t b  = ((uint64_t) hi << 32) | lo;
          k[e] = ((b < a) == keep_min) ? b : a;  // one 64-bit compare per key and step
        }
      } else {  // partner in wave ^ (j / (64 E)), same lane and register
        const int jw     = j / (64 * E);
        const bool upper = (wave & jw) != 0;
        const bool asc   = ((vlane * E) & size) == 0;
        const bool keep_min = asc != upper;
#pragma unroll
        for (int e = 0; e < E; ++e) sx[(wave * E + e) * 64 + lane] = k[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint64_t a = k[e], b = sx[((wave ^ jw) * E + e) * 64 + lane];
          k[e] = ((b < a) == keep_min) ? b : a;  // one 64-bit compare per key and step
        }
        __syncthreads();
      }
    }
  }
}

template <int E, int WAVES>
__device__ __forceinline__ void sort_tile_blocked(uint64_t* __restrict__ gk, uint32_t* __restrict__ pl, int L, int lane,
    int wave, uint64_t* __restrict__ sx) {
  const int vlane = wave * 64 + lane;
  uint64_t k[E];
#pragma unroll
  for (int e = 0; e < E; ++e) k[e] = (vlane * E + e) < L ? gk[vlane * E + e] : ~0ull;
#ifndef SKGS_EXP_NOSORT
  bitonic_blocked<E, WAVES>(k, lane, wave, sx);
#endif
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = vlane * E + e;
    if (i < L) gk[i] = k[e], pl[i] = (uint32_t) k[e];
  }
}

// Lists longer than one wave sorts on its own (> 512 keys): queued in `worklist` by the kernel below, one workgroup per
// list here.  In a dense scene (trained models at 800 x 800 reach several million tile instances) MOST tiles are such
// lists, so this path matters.  Sort = the four waves sort one chunk of 64 E keys each in registers (the network of the
// short lists, no LDS, no barrier), then log2(chunks) MERGE rounds through LDS: every thread finds the start of its E
// consecutive outputs with a merge-path binary search and merges them sequentially -- O(n log n) compare work and two
// barriers per round instead of the O(n log^2 n) bitonic steps with a barrier (or a cross-wave exchange) each.
// History at R = 3.8 M (2500 lists of ~1500 keys): whole network in LDS drained by 128 workgroups 1017 us; 1024-key
// register chunks + LDS bitonic merge levels 231 us; 2048 / 4096-key register networks over four waves 400 us.
// One merge round over the LDS image `src` (sorted runs of `len` keys, pairs of runs merged): thread t produces outputs
// [t E, t E + E) (E = outputs per thread).  LAST: they go to the tile's key list / point list in global memory instead of
// `dst`.  (Merging in place -- outputs staged in registers, written back after a barrier -- halves the LDS but the staging
// array went to scratch: 894 us instead of 135 at R = 3.8 M.)
template <int E, bool LAST>
__device__ __forceinline__ void merge_round(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
    uint64_t* __restrict__ gk, uint32_t* __restrict__ pl, int len, int L, int tid) {
  const int o0   = tid * E;
  const int pair = o0 / (2 * len), d = o0 - pair * 2 * len;
  const uint64_t* A = src + pair * 2 * len;
  const uint64_t* B = A + len;
  int lo = max(0, d - len), hi = min(d, len);
  while (lo < hi) {  // merge path: a tile's keys are distinct (depth bits << 32 | Gaussian id); paddings tie freely
    const int mid = (lo + hi) >> 1;
    if (A[mid] < B[d - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  int i = lo, j = d - lo;
  uint64_t ka = i < len ? A[i] : ~0ull, kb = j < len ? B[j] : ~0ull;
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const bool ta    = (i < len) && (j >= len || ka <= kb);
    const uint64_t v = ta ? ka : kb;
    if (ta) {
      ++i;
      ka = i < len ? A[i] : ~0ull;
    } else {
      ++j;
      kb = j < len ? B[j] : ~0ull;
    }
    if (LAST) {
      if (o0 + q < L) gk[o0 + q] = v, pl[o0 + q] = (uint32_t) v;
    } else {
      dst[o0 + q] = v;
    }
  }
}

// CPW chunks of 512 keys per wave: lists of up to 2048 (CPW = 1, 32 KB of LDS), 4096 (CPW = 2, 64 KB) or 8192 keys (CPW = 4,
// 128 KB: one workgroup per CU).  Dynamic LDS: two images of NMAX keys.
template <int CPW>
__global__ void __launch_bounds__(SORT_THREADS) tile_sort_merge_kernel(const GeomHeader* __restrict__ hdr,
    const uint32_t* __restrict__ worklist, const uint32_t* __restrict__ tile_begin, const uint32_t* __restrict__ tile_end,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list, int64_t capacity, int len_lo, int len_hi) {
  constexpr int E = 8, CH = 64 * E, NMAX = 4 * CPW * CH, OPT = NMAX / SORT_THREADS;  // OPT outputs per thread and round
  extern __shared__ __attribute__((aligned(16))) uint64_t s_sort[];
  uint64_t* s_a = s_sort;
  uint64_t* s_b = s_sort + NMAX;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nbig = hdr->big_tiles;  // sparse scenes: 0
  for (int w = blockIdx.x; w < nbig; w += gridDim.x) {
    const int tile    = (int) worklist[w];
    const int64_t s64 = tile_begin[tile], e64 = min<int64_t>((int64_t) tile_end[tile], capacity);
    const int L = (int) (e64 - s64);
    if (L <= len_lo || L > len_hi) continue;  // another instantiation's list (workgroup-uniform)
    uint64_t* gk = keys + s64;
    uint32_t* pl = point_list + s64;
    if (L > NMAX) {  // beyond the LDS window (only the widest instantiation gets here): the network on global memory
      int n = 1;
      while (n < L) n <<= 1;
      bitonic_any(gk, L, n, tid);
      for (int i = tid; i < L; i += SORT_THREADS) pl[i] = (uint32_t) gk[i];
      __syncthreads();
      continue;
    }
    // ---- phase 1: every wave sorts its chunk(s) of 512 keys in registers (all-padding chunks skip the network)
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      uint64_t k[E];
      const int base = (c * 4 + wave) * CH + lane * E;
#pragma unroll
      for (int e = 0; e < E; ++e) k[e] = (base + e) < L ? gk[base + e] : ~0ull;
      if ((c * 4 + wave) * CH < L) bitonic_blocked<E, 1>(k, lane, 0, nullptr);
#pragma unroll
      for (int e = 0; e < E; ++e) s_a[base + e] = k[e];
    }
    __syncthreads();
    // ---- phase 2: merge rounds, run length 512 -> NMAX / 2; the last one writes the list and the point list
    if (L <= 2 * CH) {
      // two real chunks: their one merge is already the last round (the generic rounds below would go on merging the result
      // with all-padding runs: at 500k Gaussians @1024^2 every long list is one of these, 36.0 -> 33.7 us)
      merge_round<OPT, true>(s_a, nullptr, gk, pl, CH, L, tid);
    } else {
      uint64_t* src = s_a;
      uint64_t* dst = s_b;
#pragma unroll
      for (int len = CH; len < NMAX / 2; len *= 2) {
        merge_round<OPT, false>(src, dst, nullptr, nullptr, len, L, tid);
        __syncthreads();
        uint64_t* t = src;
        src = dst, dst = t;
      }
      merge_round<OPT, true>(src, nullptr, gk, pl, NMAX / 2, L, tid);
    }
    __syncthreads();
  }
}

// The order the blend kernels walk the tiles in (VALUBusy 71 % forward / 84 % backward at config #1 against
// 94 % at #4 -- 10 000 single-wave workgroups over 8192 wave slots with list lengths 207 mean / 424 max is a 1.2-round
// grid whose tail is whatever long tile happens to be dispatched last).  Groups of 8 consecutive tiles (the unit that
// xcd_remap keeps on one XCD / in one L2) are ranked by their total list length, heaviest first: the hardware dispatches
// workgroups in blockIdx order, so the long lists start first and the short ones fill the tail (longest-processing-
// time-first).  Ranking = one comparison pass over the group weights in LDS (rank = number of heavier groups; ties by
// id): 313 groups at 800 x 800, no barrier ladder.  Runs as workgroup 0 of the sort launch -- the counts are final since the
// scatter, nothing here depends on the sorting -- so it costs no launch and hides behind the sort.
constexpr int ORDER_MAX_GROUPS = 4096;  // 32768 tiles (e.g. 2896 x 2896); beyond: identity order
int g_tile_order_mode = 1;
__device__ void tile_order_job(int T, int bucket, int mode, const uint32_t* __restrict__ cursors,
    const uint32_t* __restrict__ tile_begin, const uint32_t* __restrict__ tile_end, uint32_t* __restrict__ group_order) {
  __shared__ uint32_t s_key[ORDER_MAX_GROUPS];
  const int G = tile_groups(T);
  if (G > ORDER_MAX_GROUPS || mode == 0) {
    for (int g = threadIdx.x; g < G; g += blockDim.x) group_order[g] = (uint32_t) g;
    return;
  }
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    uint32_t w = 0;
    for (int t = g * TILE_GROUP; t < min(T, (g + 1) * TILE_GROUP); ++t)
      w += bucket ? min(cursors[t], (uint32_t) bucket) : tile_end[t] - tile_begin[t];
    s_key[g] = (min(w, 0xfffffu) << 12) | (uint32_t) (ORDER_MAX_GROUPS - 1 - g);  // unique: heavier first, then lower id
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const uint32_t k = s_key[g];
    int rank = 0;
    for (int h = 0; h < G; ++h) rank += s_key[h] > k ? 1 : 0;  // uniform LDS address per iteration: broadcast reads
    group_order[rank] = (uint32_t) g;
  }
}

// One workgroup = 4 waves = 4 consecutive tiles: each wave sorts its own list of at most 512 keys in registers (the steps
// are latency-bound: four waves side by side); longer lists go to the worklist of the merge kernel above.  Workgroup 0 is
// the tile-order job above.
__global__ void __launch_bounds__(256) tile_sort_wave_kernel(int T, int bucket, const uint32_t* __restrict__ cursors,
    uint32_t* __restrict__ tile_begin, uint32_t* __restrict__ tile_end, uint32_t* __restrict__ worklist, GeomHeader* hdr,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list, int64_t capacity, uint32_t* __restrict__ group_order,
    int order_mode) {
  if (blockIdx.x == 0) {
    tile_order_job(T, bucket, order_mode, cursors, tile_begin, tile_end, group_order);
    return;
  }
  __shared__ int s_big[4];
  __shared__ int s_base;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = (blockIdx.x - 1) * 4 + wave;
  int64_t s64 = 0, e64 = 0;
  if (tile < T) {
    if (bucket) {  // bucket layout: the per-tile cursor is the count; this kernel publishes the tile's range
      const uint32_t cnt = cursors[tile];
      s64 = (int64_t) tile * bucket, e64 = s64 + min(cnt, (uint32_t) bucket);
      if (lane == 0) {
        tile_begin[tile] = (uint32_t) s64, tile_end[tile] = (uint32_t) e64;
        if (cnt > (uint32_t) bucket) hdr->overflow = 1, atomicAdd(&hdr->overflow_events, 1);
      }
    } else {
      s64 = tile_begin[tile], e64 = min<int64_t>((int64_t) tile_end[tile], capacity);
    }
  }
  const int L = max((int) (e64 - s64), 0);
  // lists beyond a wave's reach go to the worklist of tile_sort_merge_kernel: ONE global atomic per workgroup (in a dense
  // scene nearly every tile is such a list: 2500 same-address atomics serialise at the memory side)
  const bool big = L > WSORT_WAVE_MAX;
  if (lane == 0) s_big[wave] = big ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = s_big[0] + s_big[1] + s_big[2] + s_big[3];
    s_base = n ? atomicAdd(&hdr->big_tiles, n) : 0;
  }
  __syncthreads();
  if (big) {
    int slot = s_base;
    for (int q = 0; q < wave; ++q) slot += s_big[q];
    if (lane == 0) worklist[slot] = (uint32_t) tile;
    return;
  }
  if (tile >= T) return;
  uint64_t* gk = keys + s64;
  uint32_t* pl = point_list + s64;
  if (L <= 1) {
    if (L == 1 && lane == 0) pl[0] = (uint32_t) gk[0];
  } else if (L <= 64) {
    sort_tile_blocked<1, 1>(gk, pl, L, lane, 0, nullptr);
  } else if (L <= 128) {
    sort_tile_blocked<2, 1>(gk, pl, L, lane, 0, nullptr);
  } else if (L <= 256) {
    sort_tile_blocked<4, 1>(gk, pl, L, lane, 0, nullptr);
  } else {
    sort_tile_blocked<8, 1>(gk, pl, L, lane, 0, nullptr);
  }
}

}  // namespace

// 1 (default): the blend kernels walk the tile groups heaviest first; 0: in raster order (A/B measurements)
extern "C" void skgs_set_tile_order(int mode) { g_tile_order_mode = mode ? 1 : 0; }

int launch_scan_tiles(GeomView g, ImgView im, int64_t P, hipStream_t s) {
  ProfScope prof(K_SCAN, s);
  if (P > 0) {
    const int64_t lanes = P * LPG;
    if (im.T <= BIN_LDS_TILES)
      hipLaunchKernelGGL(count_tiles_lds_kernel, dim3(bin_groups()), dim3(BIN_THREADS), (size_t) im.T * 4, s, (int) P,
          im.tiles_x, im.tiles_y, im.T, g.recs, im.tile_counts);
    else
      hipLaunchKernelGGL(count_tiles_kernel, dim3((unsigned) ((lanes + 255) / 256)), dim3(256), 0, s, (int) P, im.tiles_x,
          im.tiles_y, g.recs, im.tile_counts);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, im.T, im.tile_counts, im.tile_offsets,
      im.cursors, im.tile_begin, im.tile_end, g.hdr);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

// Lanes per Gaussian of the scatter launch.  The kernel is a chain of dependent memory round trips -- 73 % of its wave time is
// spent in s_waitcnt (tools/pmc_kernel.sh) -- and every grid-stride iteration of its two passes is one of them: P * LPG lanes
// over 256 x 1024 threads are 6 iterations per pass at 100k Gaussians with 16 lanes each, 1.5 with 4.  Sixteen lanes pay only
// when splats cover many tiles (the inner walk of a splat is LPG-strided); the capacity of the tile lists per Gaussian bounds
// the average number of tiles a Gaussian touches.  Measured (scatter launch, config #1 / #3 / #4): 21.7 / 42.0 / 65.1 us with 16
// lanes, 16.1 / 27.2 / 42.7 with 4 (2: the same, 1: 20.2 at #1); dense scenes (x4 scales, 38 tiles per Gaussian) do not care.
// SKGS_SCATTER_LPG overrides (4 / 8 / 16).
static int scatter_lanes(int P, int T, int bucket, int64_t capacity, int hint) {
  static const int forced = [] {
    const char* e = getenv("SKGS_SCATTER_LPG");
    return e ? atoi(e) : 0;
  }();
  if (forced == 4 || forced == 8 || forced == 16) return forced;
  // (the caller's measurement of tiles per Gaussian, if it has one: skgs_raster_inputs.tiles_per_gaussian_hint)
  const double slots_per_gaussian = hint > 0 ? (double) hint : (bucket > 0 ? (double) T * bucket : (double) capacity) / (double) (P > 0 ? P : 1);
  return slots_per_gaussian <= 24.0 ? 4 : slots_per_gaussian <= 64.0 ? 8 : 16;
}

int launch_scatter_sort(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, hipStream_t s) {
  const int P = in.P;
  const int bucket = in.tile_bucket_capacity > 0 ? in.tile_bucket_capacity : 0;
  // P == 0 (an empty scene through the C ABI): nothing to scatter, but the sort launch still runs over the T empty lists --
  // it is the launch that publishes tile_begin / tile_end in the bucket layout and the blend kernels' group order
  // (tile_order_job); skipping it left the blend launch indexing tiles through uninitialised words.
  if (P > 0) {
    ProfScope prof(K_SCATTER, s);
    const int64_t lanes = (int64_t) P * LPG;
    if (im.T <= BIN_LDS_TILES) {
#define SKGS_SCATTER(L)                                                                                                   \
  hipLaunchKernelGGL(scatter_lds_kernel<L>, dim3(bin_groups()), dim3(BIN_THREADS), (size_t) im.T * 8, s, P, im.tiles_x, \
      im.tiles_y, im.T, g.recs, im.tile_offsets, im.cursors, b.keys, b.capacity, g.hdr, bucket)
      switch (scatter_lanes(P, im.T, bucket, b.capacity, in.tiles_per_gaussian_hint)) {
        case 4: SKGS_SCATTER(4); break;
        case 8: SKGS_SCATTER(8); break;
        default: SKGS_SCATTER(16); break;
      }
#undef SKGS_SCATTER
    } else
      hipLaunchKernelGGL(scatter_kernel, dim3((unsigned) ((lanes + 255) / 256)), dim3(256), 0, s, P, im.tiles_x, im.tiles_y,
          g.recs, im.tile_offsets, im.cursors, b.keys, b.capacity, g.hdr, bucket);
  }
  SKGS_CHECK_HIP(hipGetLastError());
  {
    ProfScope prof(K_SORT, s);
    hipLaunchKernelGGL(tile_sort_wave_kernel, dim3((im.T + 3) / 4 + 1), dim3(256), 0, s, im.T, bucket, im.cursors, im.tile_begin,
        im.tile_end, im.worklist, g.hdr, b.keys, b.point_list, b.capacity, im.group_order, g_tile_order_mode);
    // lists longer than 512 keys: one workgroup per list drains the worklist (chunk sorts in registers + merge rounds in
    // LDS).  A bucket layout whose buckets hold no more than a wave sorts cannot produce one: no launch.  Lists of up to
    // 2048 keys take the one-chunk-per-wave instantiation (32 KB of LDS), up to 4096 the two-chunk one (64 KB), up to 8192
    // the four-chunk one (128 KB); beyond that the network on global memory inside the widest instantiation.
    const int longest = P == 0 ? 0 : in.longest_list_hint > 0 ? in.longest_list_hint : 0x7fffffff;  // (an upper bound, if the caller has one)
    if ((bucket == 0 || bucket > WSORT_WAVE_MAX) && longest > WSORT_WAVE_MAX)
      hipLaunchKernelGGL(tile_sort_merge_kernel<1>, dim3(std::min(im.T, 2048)), dim3(SORT_THREADS), 2 * 2048 * 8, s, g.hdr,
          im.worklist, im.tile_begin, im.tile_end, b.keys, b.point_list, b.capacity, WSORT_WAVE_MAX, 2048);
    if ((bucket == 0 || bucket > 2048) && longest > 2048)
      hipLaunchKernelGGL(tile_sort_merge_kernel<2>, dim3(std::min(im.T, 1024)), dim3(SORT_THREADS), 2 * 4096 * 8, s, g.hdr,
          im.worklist, im.tile_begin, im.tile_end, b.keys, b.point_list, b.capacity, 2048, 4096);
    if ((bucket == 0 || bucket > 4096) && longest > 4096) {
      // per launch, not once per process: the attribute is per DEVICE (and a static flag is not thread-safe)
      SKGS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_sort_merge_kernel<4>),
          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 * 8));
      hipLaunchKernelGGL(tile_sort_merge_kernel<4>, dim3(std::min(im.T, 512)), dim3(SORT_THREADS), 2 * 8192 * 8, s, g.hdr,
          im.worklist, im.tile_begin, im.tile_end, b.keys, b.point_list, b.capacity, 4096, 0x7fffffff);
    }
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
