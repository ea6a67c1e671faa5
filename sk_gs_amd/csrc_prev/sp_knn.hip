// sp_knn.hip -- calc_LBS_weight of the SUPERPOINT stage (networks/sk_gs.py:751-774 as sp_stage calls it, :844): the K nearest of
// M = 512 superpoints in 3 + 8 dimensions ([xyz | hyper_feature], the positions detached, :753-755) and one of the four
// weightings, in ONE launch; and its backward.
//
// The search is P x M x 11 subtract / multiply / add triples with a 5-slot insertion behind each: 100k x 512 candidates are
// 2.8 G lane-operations if done blindly.  Exact arithmetic (the distance is the oracle's left-to-right sum of squares, no
// contraction: indices must be BIT-exact, ties to the lower id as pytorch3d) leaves two exact savings, both wave-wide:
//   * a candidate whose xyz part alone is already no better than the last entry of the lane's list cannot enter the list
//     (adding non-negative terms never decreases an fp32 sum): when that holds for ALL 64 lanes the eight hyper dimensions are
//     skipped;
//   * when no lane's full distance beats its K-th, the insertion network is skipped.
// Superpoints live in LDS as 12-float rows (three broadcast ds_read_b128 per candidate).  The list is kept sorted with
// v_min / v_med3 on the distances (slot k becomes med3(d[k-1], d[k], cand)) and two selects per slot on the ids: 4 K
// operations instead of the 5 K of a compare-and-swap bubble.
// Weightings exactly as csrc/deform.hip::knn_dist_weights_kernel / lbs_weights_forward_kernel evaluate them (the operator
// path), so the fused step and the autograd path see the same weights bit for bit.
//
// Backward (distance-based weightings): g_weights [P,K] -> g_dist -> hyper_feature.grad [P,F] (per Gaussian) and, summed over
// the (Gaussian, neighbour) pairs that picked superpoint j, sp_hyper_feature.grad [M,F], _sp_radius.grad, _sp_weight.grad:
// LDS accumulators per workgroup, partials, fixed-order reduction (as dist_weights_backward_kernel).  The `W` weighting's
// dense [P,M] logit gradient is skgs_lbs_weights_backward.
#pragma clang fp contract(off)
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int SPK_THREADS = 512;
constexpr int CROW        = 12;  // LDS row of a superpoint: xyz, 8 hyper coordinates, pad
constexpr int MAXF        = 8;

// Per-lane select on an explicit lane mask: v_cndmask_b32 in its VOP3 form with an SGPR-pair mask.  hipcc picks the VOP2 form
// (implicit VCC) for about a third of the selects of an insertion network, and that encoding issues in 22.6 clocks per wave
// instruction on gfx950 against 4.4 for this one (tools/micro/valu_issue_rate.hip, profiles/r03_l_valu_issue_rate.txt).
__device__ __forceinline__ float sel(uint64_t m, float t, float f) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}
__device__ __forceinline__ int sel(uint64_t m, int t, int f) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}

// insertion by (distance, id) lexicographically: ties to the lower id -- the order a serial scan in id order produces, whatever
// order the candidates arrive in (the scan order, the merge of the four partial lists of a Gaussian)
template <int KCAP>
__device__ __forceinline__ void topk_insert_lex(float (&bd)[KCAP], int (&bi)[KCAP], float d, int id) {
  uint64_t lt[KCAP];
#pragma unroll
  for (int k = 0; k < KCAP; ++k)
    lt[k] = __builtin_amdgcn_ballot_w64(d < bd[k]) | (__builtin_amdgcn_ballot_w64(d == bd[k]) & __builtin_amdgcn_ballot_w64(id < bi[k]));
#pragma unroll
  for (int k = KCAP - 1; k >= 1; --k) {
    bi[k] = sel(lt[k - 1], bi[k - 1], sel(lt[k], id, bi[k]));
    bd[k] = sel(lt[k - 1], bd[k - 1], sel(lt[k], d, bd[k]));
  }
  bi[0] = sel(lt[0], id, bi[0]);
  bd[0] = sel(lt[0], d, bd[0]);
}

// the weighting's per-superpoint parameters arrive RAW (`_sp_radius`, `_sp_weight`: exp / sigmoid of sk_gs.py:547-553 applied
// here, the backward returns the raw parameters' gradients -- what the fused step hands over) or ACTIVATED (what the reference's
// calc_LBS_weight receives, sk_gs.py:759-766: the operator path)
__device__ __forceinline__ float act_radius(const float* __restrict__ r, int j, int activated) { return activated ? r[j] : expf(r[j]); }
__device__ __forceinline__ float act_kweight(const float* __restrict__ w, int j, int activated) {
  return activated ? w[j] : 1.0f / (1.0f + expf(-w[j]));
}

// One small launch in front of the search (when the caller gave the pair buffer): the list header and counters are cleared and
// the superpoint table is packed ONCE -- rows [xyz | hyper | id] in scan order -- so that each of the search's ~1500 workgroups
// fills its LDS copy with six coalesced 16-byte loads per thread instead of 24 dependent gathers (order -> position / feature).
__global__ void __launch_bounds__(256) sp_prepare_kernel(SpPrepareJob job) {
  sp_prepare_element(job, blockIdx.x * 256 + threadIdx.x);
}

// F = number of hyper dimensions (0 or 8).  FOUR lanes per Gaussian: lane `part` scans the superpoints j = part (mod 4) --
// a quad reads four consecutive 48-byte table rows, conflict-free -- and the four sorted lists are merged by (distance, id)
// over two quad shuffles.  One lane per Gaussian left the chip with 1.5 waves per SIMD on a chain of dependent LDS reads
// (154 us at P = 1e5, M = 512); four give 6 waves per SIMD and a quarter of the chain each.
constexpr int LPG = 4;
template <int KCAP, int F>
__global__ void __launch_bounds__(SPK_THREADS) sp_knn_weights_kernel(int P, int M, int K, const float* __restrict__ points,
    const float* __restrict__ feature, const float* __restrict__ sp_points, const float* __restrict__ sp_feature,
    const float* __restrict__ radius_raw, const float* __restrict__ kweight_raw, float temperature, const float* __restrict__ sp_W,
    int64_t* __restrict__ out_idx, float* __restrict__ out_weights, float* __restrict__ out_dist, uint32_t* __restrict__ pair_counts,
    uint32_t* __restrict__ pair_lists, int pair_cap, uint32_t* __restrict__ pair_header, const int32_t* __restrict__ sp_order,
    const int32_t* __restrict__ sp_rank, const float* __restrict__ packed, int activated) {
  extern __shared__ __attribute__((aligned(16))) float s_c[];  // [M][CROW] | pair filing: cnt[M], base[M]
  uint32_t* s_cnt  = reinterpret_cast<uint32_t*>(s_c + (size_t) M * CROW);
  uint32_t* s_base = s_cnt + M;
  if (pair_counts)
    for (int i = threadIdx.x; i < M; i += SPK_THREADS) s_cnt[i] = 0u;
  // table row r = superpoint sp_order[r] (any permutation: the list is kept by (distance, id), so the scan order cannot change
  // the result); its id rides in the row's last word.  A spatial order makes the 8 candidates of a wave iteration neighbours.
  if (packed) {  // (sp_prepare_kernel built the rows)
    for (int i = threadIdx.x; i < M * CROW / 4; i += SPK_THREADS)
      reinterpret_cast<float4*>(s_c)[i] = reinterpret_cast<const float4*>(packed)[i];
  } else
  for (int i = threadIdx.x; i < M * CROW; i += SPK_THREADS) {
    const int r = i / CROW, c = i - r * CROW;
    const int j = sp_order ? sp_order[r] : r;
    float v = 0.f;
    if (c < 3)
      v = sp_points[3 * j + c];
    else if (c < 3 + F)
      v = sp_feature[(size_t) j * F + c - 3];
    else if (c == CROW - 1)
      v = __builtin_bit_cast(float, j);
    s_c[i] = v;
  }
  __syncthreads();
  const int part = threadIdx.x & (LPG - 1);
  const int n    = blockIdx.x * (SPK_THREADS / LPG) + (threadIdx.x >> 2);
  const int nn   = min(n, P - 1);  // (lanes beyond P follow the last Gaussian: the wave-wide tests stay well defined)
  const float p0 = points[3 * nn], p1 = points[3 * nn + 1], p2 = points[3 * nn + 2];
  float f[MAXF];
#pragma unroll
  for (int c = 0; c < MAXF; ++c) f[c] = (F > 0 && c < F) ? feature[(size_t) nn * F + c] : 0.f;
  float bd[KCAP];
  int bi[KCAP];
#pragma unroll
  for (int k = 0; k < KCAP; ++k) bd[k] = __builtin_inff(), bi[k] = 0x7fffffff;  // (loses every (distance, id) comparison)
  // two candidates (j, j + 4) per iteration: both first rows are requested together, ONE wave-wide test skips the pair when
  // neither xyz part can enter any lane's list (the common case once the lists have filled)
  auto tail = [&](float d, const float4& c0, const float* rowp) {
    const float4 c1 = *reinterpret_cast<const float4*>(rowp + 4);
    const float4 c2 = *reinterpret_cast<const float4*>(rowp + 8);
    const float e3 = f[0] - c0.w, e4 = f[1] - c1.x, e5 = f[2] - c1.y, e6 = f[3] - c1.z, e7 = f[4] - c1.w;
    const float e8 = f[5] - c2.x, e9 = f[6] - c2.y, e10 = f[7] - c2.z;
    d += e3 * e3;
    d += e4 * e4;
    d += e5 * e5;
    d += e6 * e6;
    d += e7 * e7;
    d += e8 * e8;
    d += e9 * e9;
    d += e10 * e10;
    return d;
  };
  // Where to start: a scan that walks a spatial order from its beginning APPROACHES the Gaussians -- every candidate beats the
  // last one and is inserted.  With `sp_rank` (the inverse of sp_order) the wave starts at the table row of the superpoint that
  // was nearest to its first Gaussian in the PREVIOUS call (out_idx still holds it; any value is a valid hint -- the result does
  // not depend on the order) and walks outwards, row s0, s0 + 1, s0 - 1, s0 + 2 ...: the lists fill with near neighbours in the
  // first iterations and the wave-wide tests skip the rest.
  int s0 = 0;
  if (sp_rank) {
    const uint32_t prev = (uint32_t) out_idx[(size_t) nn * K];
    s0 = __builtin_amdgcn_readfirstlane(sp_rank[prev % (uint32_t) M]);
  }
  // scan position pos -> table row: alternately right and left of s0 (pos even: s0 + pos / 2, odd: s0 - (pos + 1) / 2).  Lane
  // `part` takes positions part + 8 t and part + 4 + 8 t: both on its own side of s0, 4 t and 4 t + 2 rows out.
  const int dir  = (part & 1) ? -1 : 1;
  const int base = (part & 1) ? -((part + 1) >> 1) : (part >> 1);
  auto wrap = [&](int r) {  // into [0, M): r is at most M away.  Integer arithmetic only (no compare + select pairs)
    r += (r >> 31) & M;
    return (int) min((unsigned) r, (unsigned) (r - M));
  };
  // `wq`: the tightest bound any lane of the quad has on the Gaussian's K-th distance -- a part whose own list is still loose
  // (it has seen a quarter of the candidates) prunes with its partners' (the merged top-K can only be tighter than each part's)
  // `seed`: with the hint, the K neighbours of the PREVIOUS call (out_idx still holds them) give a bound before the scan starts:
  // when the K ids are distinct, K superpoints lie within the largest of their CURRENT distances (the scan's own arithmetic), so
  // the K-th distance cannot exceed it -- whatever the ids are (garbage before the first call: any K distinct rows bound it).
  // The lists then only ever see candidates within that bound: the insertion network (45 instructions, run by the whole wave
  // whenever ONE lane has a candidate) fires for the few true members instead of for everything the scan meets while the lists
  // fill.
  float seed = __builtin_inff();
  if (sp_rank && K <= 8) {
    int pid[8];
    bool distinct = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) pid[k] = k < K ? (int) ((uint32_t) out_idx[(size_t) nn * K + k] % (uint32_t) M) : -1 - k;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = a + 1; b < 8; ++b) distinct = distinct && pid[a] != pid[b];
    float far = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += LPG) {  // lane `part` takes neighbours part, part + 4
      const int k = k0 + part;
      if (k < K) {
        int id = pid[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) id = q == k ? pid[q] : id;
        const int row  = sp_rank[id];
        const float4 c = *reinterpret_cast<const float4*>(s_c + row * CROW);
        const float a0 = p0 - c.x, a1 = p1 - c.y, a2 = p2 - c.z;
        float d = a0 * a0;
        d += a1 * a1;
        d += a2 * a2;
        if (F > 0) d = tail(d, c, s_c + row * CROW);
        far = fmaxf(far, d);
      }
    }
    far = fmaxf(far, dpp_mov<0xb1>(far));
    far = fmaxf(far, dpp_mov<0x4e>(far));
    seed = distinct ? far : __builtin_inff();
  }
  float wq = seed;
  const int Mq = (M + LPG - 1) / LPG;  // candidates per lane (the last ones may fall beyond M: masked by `in`)
  const int full = (M / (2 * LPG)) * 2;  // iterations-of-one (i) that need no bounds check: i + 1 < full
  // The lane's two rows as BYTE offsets into the table, advanced by 4 rows per iteration on the lane's own side of s0 and folded
  // back into [0, 48 M) with one three-way minimum (x, x - 48 M, x + 48 M as unsigned: exactly one of them is in range) -- the
  // row index arithmetic (two wraps, two 32-bit multiplies by the row size) was 19 of the ~45 instructions of a skipped iteration
  const uint32_t M48 = (uint32_t) M * (CROW * 4);
  auto fold = [&](uint32_t x) { return min(min(x, x - M48), x + M48); };
  uint32_t oa = (uint32_t) wrap(s0 + base) * (CROW * 4), ob = (uint32_t) wrap(s0 + base + 2 * dir) * (CROW * 4);
  const uint32_t ostep = (uint32_t) (dir * 4 * CROW * 4);
  const char* tbl = reinterpret_cast<const char*>(s_c);
  auto visit = [&](int i, auto checked_c) {
    constexpr bool checked = decltype(checked_c)::value;
    const float* pa_row = reinterpret_cast<const float*>(tbl + oa);
    const float* pb_row = reinterpret_cast<const float*>(tbl + ob);
    oa = fold(oa + ostep), ob = fold(ob + ostep);
    const float4 ca = *reinterpret_cast<const float4*>(pa_row);
    const float4 cb = *reinterpret_cast<const float4*>(pb_row);
    const float a0 = p0 - ca.x, a1 = p1 - ca.y, a2 = p2 - ca.z;
    const float b0 = p0 - cb.x, b1 = p1 - cb.y, b2 = p2 - cb.z;
    float da = a0 * a0;  // (0 + t = t: the oracle's `d = 0; d += df * df` starts here)
    da += a1 * a1;
    da += a2 * a2;
    float db = b0 * b0;
    db += b1 * b1;
    db += b2 * b2;
    bool ina = true, inb = true;
    if (checked) {  // the last iterations of a table whose size is not a multiple of 8: positions beyond M are masked
      const int pa = part + LPG * i, pb = pa + LPG;
      ina = pa < M, inb = (i + 1 < Mq) && pb < M;
      da = ina ? da : __builtin_inff();
      db = inb ? db : __builtin_inff();
    }
    // exact: the sums only grow, so a pair whose xyz parts already lose cannot enter ("<=": an equal distance with a lower id
    // still displaces the list's last entry)
    if (__builtin_amdgcn_ballot_w64((da <= wq) | (db <= wq)) == 0) return;
    if (F > 0) {
      if (ina) da = tail(da, ca, pa_row);
      if (inb) db = tail(db, cb, pb_row);
    }
    // (a lane whose candidate cannot enter inserts a NaN: every comparison fails, nothing moves -- no divergent branch)
    bool any = false;
    const uint64_t ma = __builtin_amdgcn_ballot_w64(da <= wq);
    if (ma != 0) {
      topk_insert_lex<KCAP>(bd, bi, sel(ma, da, __builtin_nanf("")), __builtin_bit_cast(int, pa_row[CROW - 1]));
      any = true;
    }
    const uint64_t mb = __builtin_amdgcn_ballot_w64(db <= wq);
    if (mb != 0) {
      topk_insert_lex<KCAP>(bd, bi, sel(mb, db, __builtin_nanf("")), __builtin_bit_cast(int, pb_row[CROW - 1]));
      any = true;
    }
    if (any) {  // (wave-uniform) refresh the quad's bound: min over the four parts' last entries
      float w = bd[KCAP - 1];
      w  = fminf(w, dpp_mov<0xb1>(w));  // quad_perm:[1,0,3,2]
      wq = fminf(seed, fminf(w, dpp_mov<0x4e>(w)));  // quad_perm:[2,3,0,1]
    }
  };
  int it = 0;
  for (; it + 1 < full; it += 2) visit(it, std::false_type{});
  for (; it < Mq; it += 2) visit(it, std::true_type{});
  // ---- merge the quad's four lists: after xor 1 lanes (0,1) and (2,3) agree, after xor 2 all four
#pragma unroll
  for (int step = 1; step <= 2; step <<= 1) {
    float od[KCAP];
    int oi[KCAP];
#pragma unroll
    for (int k = 0; k < KCAP; ++k) od[k] = __shfl_xor(bd[k], step), oi[k] = __shfl_xor(bi[k], step);
#pragma unroll
    for (int k = 0; k < KCAP; ++k) topk_insert_lex<KCAP>(bd, bi, od[k], oi[k]);
  }
  const bool owner = part == 0 && n < P;  // one lane per Gaussian holds the merged list from here on
  uint32_t rank[KCAP];
  if (owner) {
    // ---- weighting (sk_gs.py:759-770), the arithmetic of deform.hip::knn_dist_weights_kernel / lbs_weights_forward_kernel
    float v[KCAP];
    float sum = 0.f;
    if (sp_W) {  // softmax of the gathered logits
      float mx = -INFINITY;
  #pragma unroll
      for (int k = 0; k < KCAP; ++k) {
        v[k] = k < K ? sp_W[(size_t) n * M + bi[k]] : -INFINITY;
        mx   = fmaxf(mx, v[k]);
      }
  #pragma unroll
      for (int k = 0; k < KCAP; ++k) {
        v[k] = k < K ? expf(v[k] - mx) : 0.f;
        sum += v[k];
      }
    } else if (radius_raw) {  // exp(-d / (2 r^2)) [* s] + 1e-7, / sum
  #pragma unroll
      for (int k = 0; k < KCAP; ++k) {
        v[k] = 0.f;
        if (k < K) {
          const float r = act_radius(radius_raw, bi[k], activated);
          float e = expf(-bd[k] / (2.f * (r * r)));
          if (kweight_raw) e = e * act_kweight(kweight_raw, bi[k], activated);
          v[k] = e + 1e-7f;
          sum += v[k];
        }
      }
    } else {  // softmax(-d / temperature)
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
        if (out_dist) out_dist[(size_t) n * K + k] = bd[k];
      }
    if (pair_counts)  // file the K pairs under their superpoints: local rank now, global slots below
#pragma unroll
      for (int k = 0; k < KCAP; ++k)
        if (k < K) rank[k] = atomicAdd(&s_cnt[bi[k]], 1u);
  }
  if (!pair_counts) return;
  // ---- inverse lists (sp_backward.hip walks them): ONE global atomic per superpoint this workgroup touched reserves its slots
  __syncthreads();
  for (int j = threadIdx.x; j < M; j += SPK_THREADS) {
    const uint32_t c = s_cnt[j];
    s_base[j] = c ? atomicAdd(&pair_counts[j], c) : 0u;
  }
  __syncthreads();
  if (owner) {
#pragma unroll
    for (int k = 0; k < KCAP; ++k)
      if (k < K) {
        const uint32_t slot = s_base[bi[k]] + rank[k];
        if (slot < (uint32_t) pair_cap)
          pair_lists[(size_t) bi[k] * pair_cap + slot] = ((uint32_t) n << 4) | (uint32_t) k;
        else if (atomicExch(&pair_header[1], 1u) == 0u)  // overflow: a superpoint with more than cap Gaussians (skgs_sp_pairs_bytes);
          atomicAdd(&pair_header[2], 1u);                // the first lane to see it in this forward counts the event (never cleared here)
      }
  }
}

// ---- backward of the distance-based weightings -----------------------------------------------------------------------------
// accumulators per superpoint: [F] hyper-feature gradient, radius, kernel weight
constexpr int SPB_MAX_BLOCKS = 256;
template <int F>
__global__ void __launch_bounds__(SPK_THREADS) sp_weights_backward_kernel(int P, int M, int K, const float* __restrict__ feature,
    const float* __restrict__ sp_feature, const float* __restrict__ radius_raw, const float* __restrict__ kweight_raw,
    float temperature, const float* __restrict__ weights, const int64_t* __restrict__ indices, const float* __restrict__ nn_dist,
    const float* __restrict__ g_weights, float* __restrict__ g_feature, float* __restrict__ partials, int activated) {
  extern __shared__ float s_acc[];  // [M][V]
  constexpr int V = F + 2;
  for (int i = threadIdx.x; i < M * V; i += SPK_THREADS) s_acc[i] = 0.f;
  __syncthreads();
  // (whole waves stay in the loop: wave_group_add merges the lanes that picked the same superpoint and needs all 64 lanes)
  for (int base = blockIdx.x * SPK_THREADS; base < P; base += gridDim.x * SPK_THREADS) {
    const bool live = base + (int) threadIdx.x < P;
    const int n     = live ? base + (int) threadIdx.x : P - 1;
    const float* w  = weights + (size_t) n * K;
    const float* gw = g_weights + (size_t) n * K;
    const float* dd = nn_dist + (size_t) n * K;
    const int64_t* ix = indices + (size_t) n * K;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += w[k] * gw[k];
    float sum = 0.f;
    if (radius_raw)  // S = sum_k v_k is not stored: recomputed with the forward's arithmetic
      for (int k = 0; k < K; ++k) {
        const float r = act_radius(radius_raw, (int) ix[k], activated);
        float e = expf(-dd[k] / (2.f * (r * r)));
        if (kweight_raw) e = e * act_kweight(kweight_raw, (int) ix[k], activated);
        sum += e + 1e-7f;
      }
    float gf[MAXF];
#pragma unroll
    for (int c = 0; c < MAXF; ++c) gf[c] = 0.f;
    float fc[MAXF];
#pragma unroll
    for (int c = 0; c < MAXF; ++c) fc[c] = (F > 0 && c < F) ? feature[(size_t) n * F + c] : 0.f;
    for (int k = 0; k < K; ++k) {
      const int j = (int) ix[k];
      float acc[V];  // this pair's contribution to superpoint j: [-2 g_d (f - sf)] (F), radius, kernel weight
#pragma unroll
      for (int c = 0; c < V; ++c) acc[c] = 0.f;
      float g_d;
      if (radius_raw) {
        const float r   = act_radius(radius_raw, j, activated);
        const float e   = expf(-dd[k] / (2.f * (r * r)));
        const float sk  = kweight_raw ? act_kweight(kweight_raw, j, activated) : 1.f;
        const float g_v = (gw[k] - dot) / sum;
        const float g_e = g_v * sk;
        g_d             = g_e * e * (-1.f / (2.f * (r * r)));
        acc[F]          = g_e * e * (dd[k] / (r * r * r));
        if (kweight_raw) acc[F + 1] = g_v * e;
      } else {
        g_d = -(w[k] * (gw[k] - dot)) / temperature;
      }
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
        if (F > 0 && c < F) {
          const float t = g_d * 2.f * (fc[c] - sp_feature[(size_t) j * F + c]);
          gf[c] += t;
          acc[c] = -t;
        }
      // (global atomics straight into one [M][V] table were measured: 534 us -- 5 M atomics on 5 k hot addresses)
      wave_group_add<V>(s_acc, V, j, acc, live);
    }
    if (g_feature && live)
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
        if (F > 0 && c < F) g_feature[(size_t) n * F + c] = gf[c];
  }
  __syncthreads();
  float* dst = partials + (size_t) blockIdx.x * M * V;
  for (int i = threadIdx.x; i < M * V; i += SPK_THREADS) dst[i] = s_acc[i];
}

__global__ void __launch_bounds__(256) sp_weights_finalize_kernel(int M, int F, int nblk, const float* __restrict__ partials,
    float* __restrict__ g_sp_feature, float* __restrict__ g_radius, float* __restrict__ g_kweight,
    const float* __restrict__ radius_raw, const float* __restrict__ kweight_raw, int activated) {
  const int V = F + 2, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * V) return;
  float s = 0.f;
  for (int b0 = 0; b0 < nblk; b0 += 16) {  // 16 loads in flight per round, summed in block order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = partials[(size_t) min(b0 + u, nblk - 1) * M * V + i];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (b0 + u < nblk) s += v[u];
  }
  const int j = i / V, c = i % V;
  if (c < F) {
    if (g_sp_feature) g_sp_feature[(size_t) j * F + c] = s;
  } else if (c == F) {  // w.r.t. the RAW parameter: d exp(x) = exp(x)
    if (g_radius) g_radius[j] = radius_raw ? (activated ? s : s * expf(radius_raw[j])) : 0.f;
  } else if (g_kweight) {  // d sigmoid(x) = s (1 - s)
    float d = 0.f;
    if (kweight_raw && activated) d = 1.f;
    else if (kweight_raw) {
      const float sg = 1.0f / (1.0f + expf(-kweight_raw[j]));
      d = sg * (1.f - sg);
    }
    g_kweight[j] = s * d;
  }
}

int backward_blocks(int P) { return std::max(1, std::min((P + SPK_THREADS - 1) / SPK_THREADS, SPB_MAX_BLOCKS)); }

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_sp_lbs_weights_workspace_bytes(int32_t P, int32_t M, int32_t F) {
  return (size_t) backward_blocks(P) * M * (F + 2) * sizeof(float);
}

int skgs_sp_lbs_weights_forward(int32_t P, int32_t M, int32_t K, int32_t F, const float* points, const float* feature,
    const float* sp_points, const float* sp_feature, const float* sp_radius_raw, const float* sp_weight_raw, float temperature,
    const float* sp_W, const int32_t* sp_order, const int32_t* sp_rank, int64_t* out_idx, float* out_weights, float* out_dist,
    void* pairs, size_t pairs_bytes, int32_t params_activated, int32_t pairs_prepared, skgs_stream_t stream) {
  SKGS_REQUIRE(!sp_rank || sp_order, "sp_lbs_weights_forward: sp_rank is the inverse of sp_order: give both");
  SKGS_REQUIRE(P >= 0 && M >= 1 && K >= 1 && K <= 16 && K <= M, "sp_lbs_weights_forward: need P >= 0, 1 <= K <= min(16, M)");
  SKGS_REQUIRE(!pairs || pairs_bytes >= sp_pairs_bytes(P > 0 ? P : 1, M, K), "sp_lbs_weights_forward: pair-list buffer too small (skgs_sp_pairs_bytes)");
  SKGS_REQUIRE(F == 0 || F == 8, "sp_lbs_weights_forward: F (hyper dimensions) must be 0 or 8");
  if (P == 0) return 0;
  SKGS_REQUIRE(points && sp_points && out_idx && out_weights && (F == 0 || (feature && sp_feature)),
      "sp_lbs_weights_forward: NULL argument");
  SKGS_REQUIRE(!(sp_weight_raw && !sp_radius_raw), "sp_lbs_weights_forward: a kernel weight needs a kernel radius");
  const size_t lds = (size_t) M * CROW * 4 + (pairs ? (size_t) M * 8 : 0);
  SKGS_REQUIRE(lds <= 64 * 1024, "sp_lbs_weights_forward: too many superpoints for the LDS table (<= 1170)");
  hipStream_t s = (hipStream_t) stream;
  SpPairsView pv{};
  if (pairs) {  // the lists start empty: whatever an earlier forward filed (with or without a backward) is dropped
    pv = sp_pairs_view(pairs, P, M, K);
    if (!pairs_prepared) {  // (else: skgs_sp_net_forward's launch did it, see skgs_sp_prepare)
      const int n_clear = 64 + (M + 63) / 64 * 64;  // header + counts (contiguous: the counts start at byte 256)
      const SpPrepareJob job{M, F, sp_points, sp_feature, sp_order, pv.header, n_clear, pv.table};
      hipLaunchKernelGGL(sp_prepare_kernel, dim3((std::max(n_clear, M * CROW) + 255) / 256), dim3(256), 0, s, job);
      SKGS_CHECK_HIP(hipGetLastError());
    }
  }
  ProfScope prof(K_SP_KNN, s);
  const int per_wg = SPK_THREADS / LPG;  // four lanes per Gaussian
  const dim3 grid((P + per_wg - 1) / per_wg), block(SPK_THREADS);
#define SKGS_SPK(KCAP_, F_)                                                                                              \
  hipLaunchKernelGGL((sp_knn_weights_kernel<KCAP_, F_>), grid, block, lds, s, P, M, K, points, feature, sp_points, sp_feature, \
      sp_radius_raw, sp_weight_raw, temperature, sp_W, out_idx, out_weights, out_dist, pv.counts, pv.lists, pv.cap, pv.header, sp_order, sp_rank, (const float*) pv.table, params_activated ? 1 : 0)
  if (F == 8) {
    if (K <= 5) SKGS_SPK(5, 8); else if (K <= 8) SKGS_SPK(8, 8); else SKGS_SPK(16, 8);
  } else {
    if (K <= 5) SKGS_SPK(5, 0); else if (K <= 8) SKGS_SPK(8, 0); else SKGS_SPK(16, 0);
  }
#undef SKGS_SPK
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_sp_lbs_weights_backward(int32_t P, int32_t M, int32_t K, int32_t F, const float* feature, const float* sp_feature,
    const float* sp_radius_raw, const float* sp_weight_raw, float temperature, const float* weights, const int64_t* indices,
    const float* nn_dist, const float* g_weights, float* g_feature, float* g_sp_feature, float* g_sp_radius, float* g_sp_weight,
    void* workspace, size_t workspace_bytes, int32_t params_activated, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && M >= 1 && K >= 1 && K <= 16, "sp_lbs_weights_backward: need P >= 0, 1 <= K <= 16");
  SKGS_REQUIRE(F == 0 || F == 8, "sp_lbs_weights_backward: F (hyper dimensions) must be 0 or 8");
  SKGS_REQUIRE(P == 0 || (weights && indices && nn_dist && g_weights && (F == 0 || (feature && sp_feature))),
      "sp_lbs_weights_backward: NULL argument");
  SKGS_REQUIRE(workspace && workspace_bytes >= skgs_sp_lbs_weights_workspace_bytes(P, M, F), "sp_lbs_weights_backward: workspace too small");
  const int V = F + 2, nblk = backward_blocks(P);
  const size_t lds = (size_t) M * V * 4;
  SKGS_REQUIRE(lds <= 64 * 1024, "sp_lbs_weights_backward: too many superpoints for the LDS accumulators");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_SP_KNN_BWD, s);
  float* partials = reinterpret_cast<float*>(workspace);
  if (F == 8)
    hipLaunchKernelGGL((sp_weights_backward_kernel<8>), dim3(nblk), dim3(SPK_THREADS), lds, s, P, M, K, feature, sp_feature,
        sp_radius_raw, sp_weight_raw, temperature, weights, indices, nn_dist, g_weights, g_feature, partials, params_activated ? 1 : 0);
  else
    hipLaunchKernelGGL((sp_weights_backward_kernel<0>), dim3(nblk), dim3(SPK_THREADS), lds, s, P, M, K, feature, sp_feature,
        sp_radius_raw, sp_weight_raw, temperature, weights, indices, nn_dist, g_weights, g_feature, partials, params_activated ? 1 : 0);
  hipLaunchKernelGGL(sp_weights_finalize_kernel, dim3((M * V + 255) / 256), dim3(256), 0, s, M, F, nblk, partials, g_sp_feature,
      g_sp_radius, g_sp_weight, sp_radius_raw, sp_weight_raw, params_activated ? 1 : 0);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
