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
//  * blockIdx -> tile mapping is XCD-aware (xcd_remap): small groups of neighbouring tiles share an L2, all XCDs get
//    the same mix of image regions.
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

// Pixel rectangle (inclusive, pixel-centre coordinates) covered by one wave: PPL 8x8 quadrants of a 16x16 tile.
struct WaveRect {
  float x0, y0, x1, y1;
};
template <int PPL>
__device__ __forceinline__ WaveRect wave_rect(int tile, int sub, int gx) {
  const int tx = (tile % gx) * TILE, ty = (tile / gx) * TILE;
  WaveRect r;
  if (PPL == 4) {
    r.x0 = (float) tx, r.y0 = (float) ty, r.x1 = (float) (tx + 15), r.y1 = (float) (ty + 15);
  } else if (PPL == 2) {  // quadrants 2*sub, 2*sub+1: one 16x8 strip
    r.x0 = (float) tx, r.y0 = (float) (ty + sub * 8), r.x1 = (float) (tx + 15), r.y1 = (float) (ty + sub * 8 + 7);
  } else {
    r.x0 = (float) (tx + (sub & 1) * 8), r.y0 = (float) (ty + (sub >> 1) * 8);
    r.x1 = r.x0 + 7.f, r.y1 = r.y0 + 7.f;
  }
  return r;
}
// false only if NO pixel of the rectangle can get alpha >= 1/255 from this splat.
// alpha >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 o) =: qmax (record slot, preprocess.hip).  The
// minimum of the convex q over the rectangle (in centre-relative coordinates) is 0 if the centre is inside, else it
// lies on a face visible from the centre: the nearer vertical and/or horizontal edge, where q is a 1-D parabola.
// Every pair skipped is a pair the reference `continue`s on: the rectangle is taken in the kernel's own rounded
// differences (fl is monotonic), and `slack` bounds the fp32 evaluation error of `power` anywhere in the rectangle.
__device__ __forceinline__ bool splat_reaches_rect(float cx, float cy, float qa, float qb, float qc, float qmax,
                                                   const WaveRect& r) {
  const float X0 = r.x0 - cx, X1 = r.x1 - cx, Y0 = r.y0 - cy, Y1 = r.y1 - cy;
  const bool inx = X0 <= 0.f && X1 >= 0.f, iny = Y0 <= 0.f && Y1 >= 0.f;
  const float Xn = X0 > 0.f ? X0 : X1, Yn = Y0 > 0.f ? Y0 : Y1;  // nearest edges (meaningful when !inx / !iny)
  const float dyv = fminf(fmaxf(-qb * Xn * __builtin_amdgcn_rcpf(qc), Y0), Y1);
  const float dxh = fminf(fmaxf(-qb * Yn * __builtin_amdgcn_rcpf(qa), X0), X1);
  const float qv  = qa * Xn * Xn + 2.f * qb * Xn * dyv + qc * dyv * dyv;
  const float qh  = qa * dxh * dxh + 2.f * qb * dxh * Yn + qc * Yn * Yn;
  float qmin      = fminf(inx ? qh : qv, iny ? qv : qh);
  if (inx && iny) qmin = 0.f;
  const float mx = fmaxf(fabsf(X0), fabsf(X1)), my = fmaxf(fabsf(Y0), fabsf(Y1));
  const float slack = 4e-6f * (qa * mx * mx + qc * my * my + 2.f * fabsf(qb) * mx * my);
  return !(qmin > qmax + slack) || !(qa > 0.f && qc > 0.f);  // NaN / degenerate conic: never skip
}

// Reproducible exp for the strict (parity) build: every step is an IEEE double multiply or add, written one
// operation per statement and compiled without contraction, so the CPU oracle (exp_mode = 1) gets the same bits.
__device__ __forceinline__ float skgs_exp_strict(float x) {
  const double xd = (double) x;
  const double t  = xd * 1.4426950408889634;
  const double n  = rint(t);
  const double a  = n * 0.6931471803691238;
  const double b  = n * 1.9082149292705877e-10;
  double r        = xd - a;
  r               = r - b;
  double p        = r * (1.0 / 5040.0);
  p               = p + (1.0 / 720.0);
  p               = p * r;
  p               = p + (1.0 / 120.0);
  p               = p * r;
  p               = p + (1.0 / 24.0);
  p               = p * r;
  p               = p + (1.0 / 6.0);
  p               = p * r;
  p               = p + 0.5;
  p               = p * r;
  p               = p + 1.0;
  p               = p * r;
  p               = p + 1.0;
  return (float) ldexp(p, (int) n);
}

#pragma clang fp contract(off)
#define SKGS_STRICT 1
#define SKGS_BLEND_NS blend_strict
#include "render_blend.inl"
#undef SKGS_STRICT
#undef SKGS_BLEND_NS
#pragma clang fp contract(fast)
#define SKGS_STRICT 0
#define SKGS_BLEND_NS blend_fast
#include "render_blend.inl"
#undef SKGS_STRICT
#undef SKGS_BLEND_NS

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

__global__ void __launch_bounds__(256) extra_forward_kernel(int W, int H, int gx, int E, TileRanges ranges,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ point_extra, float* __restrict__ pixel_extra) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = ranges.begin[tile], end = min<int64_t>((int64_t) ranges.end[tile], capacity);
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

__global__ void __launch_bounds__(256) extra_backward_kernel(int W, int H, int gx, int E, TileRanges ranges,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ out_opacity, const uint32_t* __restrict__ n_contrib, const float* __restrict__ point_extra,
    const float* __restrict__ dL_dpixel_extra, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dpoint_extra) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = ranges.begin[tile], end = min<int64_t>((int64_t) ranges.end[tile], capacity);
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

__global__ void __launch_bounds__(256) topk_kernel(int topk, int W, int H, int gx, TileRanges ranges,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const uint32_t* __restrict__ n_contrib, int32_t* __restrict__ top_indices, float* __restrict__ top_weights) {
  const int tile = blockIdx.x;
  const int px = (tile % gx) * TILE + (threadIdx.x & 15), py = (tile / gx) * TILE + (threadIdx.x >> 4);
  if (!(px < W && py < H)) return;
  const uint32_t pid = (uint32_t) W * py + px;
  const int64_t start = ranges.begin[tile], end = min<int64_t>((int64_t) ranges.end[tile], capacity);
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

// pixels per lane: 1 everywhere.  Measured (tools/ppl_sweep.py, fwd / bwd us): 500k Gaussians @1024^2: 171/461 (1),
// 278/564 (2), 380/702 (4); 200k @512^2: 76/186, 128/305, 253/586; 300k @1600x1200: 167/499, 242/544, 286/603.
// The 2- and 4-pixel variants stay for skgs_set_pixels_per_lane() experiments and as test cases.
inline int choose_ppl(int /*T*/) { return 1; }
int g_ppl_override = 0;
int g_strict        = 0;

}  // namespace

extern "C" void skgs_set_pixels_per_lane(int ppl) { g_ppl_override = (ppl == 1 || ppl == 2 || ppl == 4) ? ppl : 0; }
extern "C" void skgs_set_strict_math(int on) { g_strict = on ? 1 : 0; }
// layout of slots 0..4 of the gradient rows written by render_backward (render_blend.inl): moments in the fast build
bool gradacc_rows_hold_moments() { return g_strict == 0; }

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
  ProfScope prof(K_RENDER_FWD, s);
#define FWD(E_, PPL_)                                                                                                   \
  {                                                                                                                     \
    const int nblk = xcd_grid(im.T * (4 / PPL_));                                                                       \
    if (g_strict)                                                                                                       \
      hipLaunchKernelGGL((blend_strict::render_forward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, \
          im.T, TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, in.extras, in.background, im.n_contrib, out_color,   \
          out_opacity, out_extra, (uint32_t*) nullptr);                                                                 \
    else                                                                                                                \
      hipLaunchKernelGGL((blend_fast::render_forward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x,   \
          im.T, TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, in.extras, in.background, im.n_contrib, out_color,   \
          out_opacity, out_extra, (uint32_t*) nullptr);                                                                 \
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

// parity tests only: the product forward walk (one pixel per lane, no extras; strict or fast as set) with the census
// fingerprint of every pixel's blended list entries
int launch_render_census(int W, int H, GeomView g, ImgView im, BinView b, float* out_color, float* out_opacity,
    uint32_t* census, hipStream_t s) {
  const int nblk = xcd_grid(im.T * 4);
  if (g_strict)
    hipLaunchKernelGGL((blend_strict::render_forward_kernel<1, 0, true>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, im.T,
        TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, (const float*) nullptr,
        (const float*) nullptr, im.n_contrib, out_color, out_opacity, (float*) nullptr, census);
  else
    hipLaunchKernelGGL((blend_fast::render_forward_kernel<1, 0, true>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, im.T,
        TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, (const float*) nullptr,
        (const float*) nullptr, im.n_contrib, out_color, out_opacity, (float*) nullptr, census);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_backward(const skgs_raster_inputs& in, GeomView g, ImgView im, BinView b, const float* out_opacity,
    const float* dL_dcolor, const float* dL_dopacity, const float* dL_dextra, float* gradacc, hipStream_t s) {
  const int W = in.image_width, H = in.image_height;
  const int E = (in.extras && dL_dextra) ? in.E : 0;
  // (A two-pixels-per-lane variant written on packed fp32 vectors was measured at 245 us vs 225 us for this one on
  // config #1: 0.67x the visits but ~1.5x the instructions per visit once the compiler's register shuffling for the
  // v_pk_* operands is counted.)
  const int ppl = g_ppl_override ? g_ppl_override : choose_ppl(im.T);
  SKGS_REQUIRE(in.P <= (1 << 26), "rasterize_backward: the gradient rows are addressed with 32-bit byte offsets (P <= 2^26)");
  ProfScope prof(K_RENDER_BWD, s);
#define BWD(E_, PPL_)                                                                                                  \
  {                                                                                                                    \
    const int nblk = xcd_grid(im.T * (4 / PPL_));                                                                      \
    if (g_strict)                                                                                                      \
      hipLaunchKernelGGL((blend_strict::render_backward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H,           \
 im.tiles_x, im.T, TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, in.extras, in.background, out_opacity,       \
          im.n_contrib, dL_dcolor, dL_dextra, dL_dopacity, gradacc);                                                               \
    else                                                                                                               \
      hipLaunchKernelGGL((blend_fast::render_backward_kernel<PPL_, E_>), dim3(nblk), dim3(64), 0, s, W, H, im.tiles_x, \
          im.T, TileRanges{im.tile_begin, im.tile_end, im.group_order}, b.capacity, b.point_list, g.recs, in.extras, in.background, out_opacity, im.n_contrib,   \
          dL_dcolor, dL_dextra, dL_dopacity, gradacc);                                                                           \
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
  hipLaunchKernelGGL(extra_forward_kernel, dim3(im.T), dim3(256), 0, s, W, H, im.tiles_x, E, TileRanges{im.tile_begin, im.tile_end}, b.capacity,
      b.point_list, g.recs, im.n_contrib, extra, pixel_extra);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_extra_backward(int W, int H, int /*P*/, int E, const float* extra, const float* out_opacity,
    const float* grad_pixel_extra, GeomView g, ImgView im, BinView b, float* grad_means2D, float* grad_conic,
    float* grad_opacity, float* dL_dextra, hipStream_t s) {
  hipLaunchKernelGGL(extra_backward_kernel, dim3(im.T), dim3(256), 0, s, W, H, im.tiles_x, E, TileRanges{im.tile_begin, im.tile_end}, b.capacity,
      b.point_list, g.recs, out_opacity, im.n_contrib, extra, grad_pixel_extra, grad_means2D, grad_conic, grad_opacity,
      dL_dextra);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_topk(int topk, int W, int H, GeomView g, ImgView im, BinView b, int32_t* top_idx, float* top_w, hipStream_t s) {
  hipLaunchKernelGGL(topk_kernel, dim3(im.T), dim3(256), 0, s, topk, W, H, im.tiles_x, TileRanges{im.tile_begin, im.tile_end}, b.capacity,
      b.point_list, g.recs, im.n_contrib, top_idx, top_w);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
