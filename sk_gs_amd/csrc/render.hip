// render.hip -- tile alpha-blend forward and backward (gfx950), plus the optional extra-feature and top-k passes.
//
// Reference: renderCUDA_forward / renderCUDA_backward (gaussian_render.cu:16-112,182-341): one 256-thread block per
// 16x16 tile, one thread per pixel, 256-entry batches staged in shared memory behind __syncthreads_count, colours
// fetched per thread from global, and the backward scatters 9 fp32 atomicAdd per (pixel, Gaussian) pair.
//
// CDNA4 design (not a translation):
//  * the unit of work is ONE 64-lane wave with its own workgroup: no block barriers, every wave walks its tile's
//    list at its own pace and retires as soon as ITS pixels are saturated;
//  * a wave covers PPL 8x8 pixel quadrants of the tile (PPL = pixels per lane, 1/2/4): per-Gaussian LDS broadcast
//    reads and (in the backward) the cross-lane reduction are amortised over PPL pixels, and the PPL independent
//    per-pixel recurrences give the scheduler ILP;
//  * per-Gaussian records (48 B, written by the preprocess kernel) are gathered once per 64-entry batch by the
//    whole wave and broadcast from LDS;
//  * backward: per-pixel partials are summed lane-locally over the PPL pixels, reduced across the wave with DPP
//    row shifts/broadcasts, parked in an LDS row per contributing Gaussian, and flushed with one 64-byte-row
//    atomic per (Gaussian, wave) -- the shape the MI355X memory-side float atomic unit likes -- instead of 9
//    scattered atomics per pair;
//  * blockIdx -> tile mapping is XCD-aware (xcd_remap) so neighbouring tiles share an L2.
// Per-pixel arithmetic follows the reference's sequence (power, alpha clamp 0.99, 1/255 and 1e-4 tests, T/(1-a)
// recurrence seeded from 1 - out_opacity); exp is the hardware v_exp_f32 path.
#include "skgs_common.h"

namespace skgs {
namespace {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN     = 0.0001f;

template <int PPL>
struct Pix {
  float x[PPL], y[PPL];
  bool inside[PPL];
  uint32_t id[PPL];
};

template <int PPL>
__device__ __forceinline__ Pix<PPL> pixel_setup(int tile, int sub, int lane, int gx, int W, int H) {
  Pix<PPL> p;
  const int tx = tile % gx, ty = tile / gx;
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    const int q  = sub * PPL + i;  // 8x8 quadrant of the 16x16 tile
    const int px = tx * TILE + (q & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (lane >> 3);
    p.x[i]       = (float) px;
    p.y[i]       = (float) py;
    p.inside[i]  = px < W && py < H;
    p.id[i]      = (uint32_t) W * py + px;
  }
  return p;
}

// ====================================================================================================== forward
template <int PPL, int E>
__global__ void __launch_bounds__(64) render_forward_kernel(int W, int H, int gx, int T, const uint32_t* __restrict__ offsets,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ extra, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_opacity, float* __restrict__ out_extra) {
  constexpr int SUBS = 4 / PPL;
  const int v        = xcd_remap(blockIdx.x, T * SUBS);
  if (v >= T * SUBS) return;
  const int tile = v / SUBS, sub = v % SUBS;
  const int lane = threadIdx.x;
  const Pix<PPL> pix = pixel_setup<PPL>(tile, sub, lane, gx, W, H);

  __shared__ float4 s_a[WAVE];  // x, y, conic a, conic b
  __shared__ float4 s_b[WAVE];  // conic c, opacity, r, g
  __shared__ float s_c[WAVE];   // b
  __shared__ float s_e[E > 0 ? WAVE * E : 1];

  const int64_t start = offsets[tile];
  const int64_t end   = min<int64_t>((int64_t) offsets[tile + 1], capacity);

  float Tr[PPL], C[PPL][3], Ex[PPL][E > 0 ? E : 1];
  uint32_t last[PPL];
  bool done[PPL];
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    Tr[i] = 1.0f, last[i] = 0, done[i] = !pix.inside[i];
    C[i][0] = C[i][1] = C[i][2] = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) Ex[i][e] = 0.f;
  }

  for (int64_t base = start; base < end; base += WAVE) {
    bool all_done = true;
#pragma unroll
    for (int i = 0; i < PPL; ++i) all_done = all_done && done[i];
    if (__all(all_done)) break;
    const int n = (int) min<int64_t>(WAVE, end - base);
    __syncthreads();  // single-wave workgroup: orders the LDS reads of the previous batch before these writes
    if (lane < n) {
      const uint32_t id = point_list[base + lane];
      const float4 a = recs[3 * id], b = recs[3 * id + 1], c = recs[3 * id + 2];
      s_a[lane] = a, s_b[lane] = b, s_c[lane] = c.x;
#pragma unroll
      for (int e = 0; e < E; ++e) s_e[lane * (E > 0 ? E : 1) + e] = extra[(size_t) id * E + e];
    }
    __syncthreads();
    const uint32_t contrib0 = (uint32_t) (base - start);
    for (int j = 0; j < n; ++j) {
      const float4 a = s_a[j];
      const float4 b = s_b[j];
      bool hit[PPL];
      float wgt[PPL];
      bool any = false;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        hit[i] = false;
        wgt[i] = 0.f;
        if (!done[i]) {
          const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          if (power <= 0.0f) {
            const float alpha = fminf(0.99f, b.y * __expf(power));
            if (alpha >= ALPHA_MIN) {
              const float test_T = Tr[i] * (1.f - alpha);
              if (test_T < T_MIN) {
                done[i] = true;
              } else {
                hit[i]  = true;
                wgt[i]  = alpha * Tr[i];
                Tr[i]   = test_T;
                last[i] = contrib0 + j + 1;
              }
            }
          }
        }
        any = any || hit[i];
      }
      if (__any(any)) {
        const float cb = s_c[j];
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          // hit[i] false -> wgt 0: adds an exact +0
          C[i][0] += b.z * wgt[i];
          C[i][1] += b.w * wgt[i];
          C[i][2] += cb * wgt[i];
#pragma unroll
          for (int e = 0; e < E; ++e) Ex[i][e] += s_e[j * (E > 0 ? E : 1) + e] * wgt[i];
        }
      }
    }
  }
  const size_t HW = (size_t) H * W;
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    if (pix.inside[i]) {
      out_opacity[pix.id[i]] = 1.f - Tr[i];
      n_contrib[pix.id[i]]   = last[i];
      out_color[pix.id[i]]          = C[i][0];
      out_color[HW + pix.id[i]]     = C[i][1];
      out_color[2 * HW + pix.id[i]] = C[i][2];
#pragma unroll
      for (int e = 0; e < E; ++e) out_extra[e * HW + pix.id[i]] = Ex[i][e];
    }
  }
}

// ===================================================================================================== backward
// gradacc row layout (16 floats = 64 B per Gaussian):
//   0 mean2D.x  1 mean2D.y  2 conic.x  3 conic.y  4 conic.w  5 opacity  6..8 colour  9..12 extras  13..15 unused
template <int PPL, int E>
__global__ void __launch_bounds__(64) render_backward_kernel(int W, int H, int gx, int T, const uint32_t* __restrict__ offsets,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ extra, const float* __restrict__ out_opacity, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dout_extra,
    const float* __restrict__ dL_dout_opacity, float* __restrict__ gradacc) {
  constexpr int SUBS = 4 / PPL;
  constexpr int NV   = 9 + E;
  const int v        = xcd_remap(blockIdx.x, T * SUBS);
  if (v >= T * SUBS) return;
  const int tile = v / SUBS, sub = v % SUBS;
  const int lane = threadIdx.x;
  const Pix<PPL> pix = pixel_setup<PPL>(tile, sub, lane, gx, W, H);

  __shared__ float4 s_a[WAVE];
  __shared__ float4 s_b[WAVE];
  __shared__ float s_c[WAVE];
  __shared__ uint32_t s_id[WAVE];
  __shared__ float s_e[E > 0 ? WAVE * E : 1];
  __shared__ float s_acc[WAVE][GRAD_ROW];
  __shared__ uint32_t s_acc_id[WAVE];

  const int64_t start = offsets[tile];
  const int64_t end   = min<int64_t>((int64_t) offsets[tile + 1], capacity);
  const size_t HW     = (size_t) H * W;

  float T_final[PPL], Tr[PPL], dL_dT[PPL], dpix[PPL][3], dex[PPL][E > 0 ? E : 1];
  float accum[PPL][3], lastc[PPL][3], accum_e[PPL][E > 0 ? E : 1], last_e[PPL][E > 0 ? E : 1], last_alpha[PPL];
  uint32_t lastk[PPL];
  uint32_t maxk = 0;
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    const bool in = pix.inside[i];
    T_final[i]    = in ? 1.0f - out_opacity[pix.id[i]] : 0.f;
    Tr[i]         = T_final[i];
    dL_dT[i]      = in ? -dL_dout_opacity[pix.id[i]] : 0.f;
    lastk[i]      = in ? n_contrib[pix.id[i]] : 0u;
    maxk          = max(maxk, lastk[i]);
    last_alpha[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dpix[i][c]  = in ? dL_dpixels[c * HW + pix.id[i]] : 0.f;
      accum[i][c] = 0.f, lastc[i][c] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      dex[i][e]     = in ? dL_dout_extra[e * HW + pix.id[i]] : 0.f;
      accum_e[i][e] = 0.f, last_e[i][e] = 0.f;
    }
  }
  // wave-wide maximum of the last contributor: nothing behind it can matter to this wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) maxk = max(maxk, (uint32_t) __shfl_xor((int) maxk, d));
  if (maxk == 0) return;
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

  // walk the list back to front: entry at list position k (0-based) has "contributor" index k
  for (int64_t hi = start + (int64_t) min<int64_t>(maxk, end - start); hi > start; hi -= WAVE) {
    const int n = (int) min<int64_t>(WAVE, hi - start);
    __syncthreads();
    if (lane < n) {
      const uint32_t id = point_list[hi - 1 - lane];
      const float4 a = recs[3 * id], b = recs[3 * id + 1], c = recs[3 * id + 2];
      s_a[lane] = a, s_b[lane] = b, s_c[lane] = c.x, s_id[lane] = id;
#pragma unroll
      for (int e = 0; e < E; ++e) s_e[lane * (E > 0 ? E : 1) + e] = extra[(size_t) id * E + e];
    }
    __syncthreads();
    int nact = 0;  // wave-uniform count of LDS rows in use
    for (int j = 0; j < n; ++j) {
      const uint32_t k = (uint32_t) (hi - 1 - j - start);
      const float4 a = s_a[j];
      const float4 b = s_b[j];
      float g[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) g[q] = 0.f;
      bool any = false;
      float col[3] = {b.z, b.w, 0.f};
      bool col_loaded = false;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        if (k < lastk[i]) {
          const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          if (power <= 0.0f) {
            const float G     = __expf(power);
            const float alpha = fminf(0.99f, b.y * G);
            if (alpha >= ALPHA_MIN) {
              any = true;
              if (!col_loaded) col[2] = s_c[j], col_loaded = true;
              const float Tn = Tr[i] / (1.f - alpha);
              Tr[i]          = Tn;
              const float dchannel_dcolor = alpha * Tn;
              float dL_dalpha = 0.0f;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                accum[i][c] = last_alpha[i] * lastc[i][c] + (1.f - last_alpha[i]) * accum[i][c];
                lastc[i][c] = col[c];
                dL_dalpha += (col[c] - accum[i][c]) * dpix[i][c];
                g[6 + c] += dchannel_dcolor * dpix[i][c];
              }
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const float ce = s_e[j * (E > 0 ? E : 1) + e];
                accum_e[i][e]  = last_alpha[i] * last_e[i][e] + (1.f - last_alpha[i]) * accum_e[i][e];
                last_e[i][e]   = ce;
                dL_dalpha += (ce - accum_e[i][e]) * dex[i][e];
                g[9 + e] += dchannel_dcolor * dex[i][e];
              }
              dL_dalpha *= Tn;
              last_alpha[i] = alpha;
              dL_dalpha += (-T_final[i] / (1.f - alpha)) * dL_dT[i];
              const float dL_dG    = b.y * dL_dalpha;
              const float gdx      = G * dx;
              const float gdy      = G * dy;
              const float dG_ddelx = -gdx * a.z - gdy * a.w;
              const float dG_ddely = -gdy * b.x - gdx * a.w;
              g[0] += dL_dG * dG_ddelx * ddelx_dx;
              g[1] += dL_dG * dG_ddely * ddely_dy;
              g[2] += -0.5f * gdx * dx * dL_dG;
              g[3] += -0.5f * gdx * dy * dL_dG;
              g[4] += -0.5f * gdy * dy * dL_dG;
              g[5] += G * dL_dalpha;
            }
          }
        }
      }
      if (__any(any)) {
#pragma unroll
        for (int q = 0; q < NV; ++q) g[q] = wave_sum_to_lane63(g[q]);
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < NV; ++q) s_acc[nact][q] = g[q];
          s_acc_id[nact] = s_id[j];
        }
        ++nact;
      }
    }
    // flush: 4 rows (4 x 64-B lines) per wave-wide atomic instruction
    __syncthreads();
    for (int r0 = 0; r0 < nact; r0 += 4) {
      const int row = r0 + (lane >> 4), colx = lane & 15;
      if (row < nact && colx < NV) atomicAdd(&gradacc[(size_t) s_acc_id[row] * GRAD_ROW + colx], s_acc[row][colx]);
    }
  }
}

// ================================================================================== extra features (any E) + top-k
// Not on the training path (E = 0 there): straightforward one-lane-per-pixel kernels over the saved buffers.
// Reference: gaussian_rasterizer_extra.cu:10-220, gaussian_topk.cu:10-96.
__device__ __forceinline__ bool blend_alpha(const float4& a, const float4& b, float px, float py, float& alpha, float& G,
    float& dx, float& dy) {
  dx = a.x - px, dy = a.y - py;
  const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
  if (power > 0.0f) return false;
  G     = __expf(power);
  alpha = fminf(0.99f, b.y * G);
  return alpha >= ALPHA_MIN;
}

__global__ void __launch_bounds__(256) extra_forward_kernel(int W, int H, int gx, int E, const uint32_t* __restrict__ offsets,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ point_extra, float* __restrict__ pixel_extra) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = offsets[tile], end = min<int64_t>((int64_t) offsets[tile + 1], capacity);
  const uint32_t lastk = n_contrib[pid];
  float* out = pixel_extra + (size_t) pid * E;
  for (int es = 0; es < E; es += 16) {
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float Tr = 1.0f;
    uint32_t contributor = 0;
    for (int64_t k = start; k < end; ++k) {
      contributor++;
      if (contributor > lastk) break;
      const uint32_t id = point_list[k];
      const float4 a = recs[3 * id], b = recs[3 * id + 1];
      float alpha, G, dx, dy;
      if (!blend_alpha(a, b, (float) px, (float) py, alpha, G, dx, dy)) continue;
      const float test_T = Tr * (1.f - alpha);
      if (test_T < T_MIN) break;
      const float w = alpha * Tr;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (es + e < E) acc[e] += point_extra[(size_t) id * E + es + e] * w;
      Tr = test_T;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (es + e < E) out[es + e] = acc[e];
  }
}

__global__ void __launch_bounds__(256) extra_backward_kernel(int W, int H, int gx, int E, const uint32_t* __restrict__ offsets,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ out_opacity, const uint32_t* __restrict__ n_contrib, const float* __restrict__ point_extra,
    const float* __restrict__ dL_dpixel_extra, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dpoint_extra) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = offsets[tile], end = min<int64_t>((int64_t) offsets[tile + 1], capacity);
  const uint32_t lastk = n_contrib[pid];
  const float T_final = 1.0f - out_opacity[pid];
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  for (int es = 0; es < E; es += 16) {
    float accum[16], laste[16], dpx[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      accum[e] = 0.f, laste[e] = 0.f;
      dpx[e] = (es + e < E) ? dL_dpixel_extra[(size_t) pid * E + es + e] : 0.f;
    }
    float Tr = T_final, last_alpha = 0.f;
    for (int64_t k = min<int64_t>(end, start + lastk); k-- > start;) {
      const uint32_t id = point_list[k];
      const float4 a = recs[3 * id], b = recs[3 * id + 1];
      float alpha, G, dx, dy;
      if (!blend_alpha(a, b, (float) px, (float) py, alpha, G, dx, dy)) continue;
      Tr = Tr / (1.f - alpha);
      const float dch = alpha * Tr;
      float dL_dalpha = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (es + e < E) {
          const float c = point_extra[(size_t) id * E + es + e];
          accum[e] = last_alpha * laste[e] + (1.f - last_alpha) * accum[e];
          laste[e] = c;
          dL_dalpha += (c - accum[e]) * dpx[e];
          atomicAdd(&dL_dpoint_extra[(size_t) id * E + es + e], dch * dpx[e]);
        }
      }
      dL_dalpha *= Tr;
      last_alpha = alpha;
      const float dL_dG = b.y * dL_dalpha;
      const float gdx = G * dx, gdy = G * dy;
      const float dG_ddelx = -gdx * a.z - gdy * a.w;
      const float dG_ddely = -gdy * b.x - gdx * a.w;
      atomicAdd(&dL_dmean2D[3 * id + 0], dL_dG * dG_ddelx * ddelx_dx);
      atomicAdd(&dL_dmean2D[3 * id + 1], dL_dG * dG_ddely * ddely_dy);
      atomicAdd(&dL_dconic[4 * id + 0], -0.5f * gdx * dx * dL_dG);
      atomicAdd(&dL_dconic[4 * id + 1], -0.5f * gdx * dy * dL_dG);
      atomicAdd(&dL_dconic[4 * id + 3], -0.5f * gdy * dy * dL_dG);
      atomicAdd(&dL_dopacity[id], G * dL_dalpha);
    }
  }
}

__global__ void __launch_bounds__(256) topk_kernel(int topk, int W, int H, int gx, const uint32_t* __restrict__ offsets,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const uint32_t* __restrict__ n_contrib, int32_t* __restrict__ top_indices, float* __restrict__ top_weights) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = offsets[tile], end = min<int64_t>((int64_t) offsets[tile + 1], capacity);
  const uint32_t lastk = n_contrib[pid];
  float* tw   = top_weights + (size_t) pid * topk;
  int32_t* ti = top_indices + (size_t) pid * topk;
  for (int q = 0; q < topk; ++q) tw[q] = 0.f, ti[q] = -1;
  float Tr = 1.0f;
  uint32_t contributor = 0;
  for (int64_t k = start; k < end; ++k) {
    contributor++;
    if (contributor >= lastk) {  // reference: `continue` for every later entry (gaussian_topk.cu:61) == stop
      break;
    }
    const uint32_t id = point_list[k];
    const float4 a = recs[3 * id], b = recs[3 * id + 1];
    float alpha, G, dx, dy;
    if (!blend_alpha(a, b, (float) px, (float) py, alpha, G, dx, dy)) continue;
    const float test_T = Tr * (1.f - alpha);
    if (test_T < T_MIN) break;
    float w     = alpha * Tr;
    int32_t idx = (int32_t) id;
    for (int q = 0; q < topk; ++q) {
      if (w >= tw[q]) {
        const float t0 = tw[q];
        tw[q]          = w;
        w              = t0;
        const int32_t i0 = ti[q];
        ti[q]            = idx;
        idx              = i0;
      }
    }
    Tr = test_T;
  }
}

// pixels-per-lane heuristic: keep >= ~4 waves per SIMD in flight (1024 SIMDs on MI355X)
inline int choose_ppl(int T) {
  if (T >= 8192) return 4;
  if (T >= 3072) return 2;
  return 1;
}
int g_ppl_override = 0;

}  // namespace

extern "C" void skgs_set_pixels_per_lane(int ppl) { g_ppl_override = (ppl == 1 || ppl == 2 || ppl == 4) ? ppl : 0; }

#define SKGS_DISPATCH_E(E_, FN, ...)     \
  switch (E_) {                          \
    case 0: FN(0, __VA_ARGS__); break;   \
    case 1: FN(1, __VA_ARGS__); break;   \
    case 2: FN(2, __VA_ARGS__); break;   \
    case 3: FN(3, __VA_ARGS__); break;   \
    case 4: FN(4, __VA_ARGS__); break;   \
    default: return set_error("Only Support 0,1,2,3,4 extra features (got %d)", E_); \
  }

int launch_render_forward(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, float* out_color,
    float* out_opacity, float* out_extra, hipStream_t s) {
  const int W = in.image_width, H = in.image_height;
  const int E = in.extras ? in.E : 0;
  const int ppl = g_ppl_override ? g_ppl_override : choose_ppl(im.T);
#define FWD(E_, PPL_)                                                                                                   \
  {                                                                                                                     \
    const int nblk = ((im.T * (4 / PPL_) + 7) / 8) * 8;                                                                 \
    hipLaunchKernelGGL((render_forward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, im.T,           \
        im.tile_offsets, b.capacity, b.point_list, g.recs, in.extras, im.n_contrib, out_color, out_opacity, out_extra); \
  }
  if (ppl == 4) {
    SKGS_DISPATCH_E(E, FWD, 4)
  } else if (ppl == 2) {
    SKGS_DISPATCH_E(E, FWD, 2)
  } else {
    SKGS_DISPATCH_E(E, FWD, 1)
  }
#undef FWD
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_backward(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, const float* out_opacity,
    const float* dL_dcolor, const float* dL_dopacity, const float* dL_dextra, float* gradacc, hipStream_t s) {
  const int W = in.image_width, H = in.image_height;
  const int E = (in.extras && dL_dextra) ? in.E : 0;
  const int ppl = g_ppl_override ? g_ppl_override : choose_ppl(im.T);
#define BWD(E_, PPL_)                                                                                                  \
  {                                                                                                                    \
    const int nblk = ((im.T * (4 / PPL_) + 7) / 8) * 8;                                                                \
    hipLaunchKernelGGL((render_backward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, im.T,         \
        im.tile_offsets, b.capacity, b.point_list, g.recs, in.extras, out_opacity, im.n_contrib, dL_dcolor, dL_dextra, \
        dL_dopacity, gradacc);                                                                                         \
  }
  if (ppl == 4) {
    SKGS_DISPATCH_E(E, BWD, 4)
  } else if (ppl == 2) {
    SKGS_DISPATCH_E(E, BWD, 2)
  } else {
    SKGS_DISPATCH_E(E, BWD, 1)
  }
#undef BWD
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_extra_forward(int W, int H, int /*P*/, int E, const float* extra, GeomView g, ImgView im, BinView b,
    float* pixel_extra, hipStream_t s) {
  hipLaunchKernelGGL(extra_forward_kernel, dim3(im.T), dim3(256), 0, s, W, H, im.tiles_x, E, im.tile_offsets, b.capacity,
      b.point_list, g.recs, im.n_contrib, extra, pixel_extra);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_extra_backward(int W, int H, int /*P*/, int E, const float* extra, const float* out_opacity,
    const float* grad_pixel_extra, GeomView g, ImgView im, BinView b, float* grad_means2D, float* grad_conic,
    float* grad_opacity, float* dL_dextra, hipStream_t s) {
  hipLaunchKernelGGL(extra_backward_kernel, dim3(im.T), dim3(256), 0, s, W, H, im.tiles_x, E, im.tile_offsets, b.capacity,
      b.point_list, g.recs, out_opacity, im.n_contrib, extra, grad_pixel_extra, grad_means2D, grad_conic, grad_opacity,
      dL_dextra);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_topk(int topk, int W, int H, GeomView g, ImgView im, BinView b, int32_t* top_idx, float* top_w, hipStream_t s) {
  hipLaunchKernelGGL(topk_kernel, dim3(im.T), dim3(256), 0, s, topk, W, H, im.tiles_x, im.tile_offsets, b.capacity,
      b.point_list, g.recs, im.n_contrib, top_idx, top_w);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
