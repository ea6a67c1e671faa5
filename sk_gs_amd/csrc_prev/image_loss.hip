// image_loss.hip -- fused training-image loss  lambda_l1 * mean|x - y| + lambda_ssim * (1 - mean SSIM(x, y))
// forward and backward (gfx950).  Scope-table row (f)-1: "fused image loss".
//
// Reference: networks/losses/ssim.py:20-62 (11x11 Gaussian window sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2,
// five depth-wise conv2d + ~15 element-wise torch kernels forward, as many again backward), networks/losses/
// image_loss.py:6-32 (L1 mean), weights exps/default.yaml:83-84.  On MI355X the depth-wise convs run in MIOpen at
// ~0.4 ms each -- 3.5 ms per training step, more than the whole rasterizer.
//
// Here: the window is separable.  Forward, per 32x32 output tile: the 11-tap horizontal pass runs straight out of global
// memory (a thread slides the window over 18 inputs of x and y for 8 outputs) for FOUR moments -- E[x], E[y], E[xx + yy],
// E[xy]: SSIM needs the two variances only as their sum -- into LDS, the vertical pass out of LDS into registers (14
// filtered rows for 4 outputs), then the SSIM map and its partial derivatives w.r.t. (mu1, E[xx], E[xy]) in place (one
// division per pixel), and the tile's SSIM and L1 sums.  Backward = the same separable convolution applied to the three
// derivative maps (staged in LDS once, coalesced):
//     dL/dx = gs * (w * dmu1 + 2 x (w * dExx) + y (w * dExy)) + gl * sign(x - y).
// HBM traffic: forward reads 2 and writes 3 image planes, backward reads 5 and writes 1 (vs ~60 plane passes in
// the torch graph).  Per-tile partial sums are reduced in a fixed order: the loss value is bitwise reproducible.
// Round 3 (tools/time_loss.py, 800x800, graph of 20): 21.8 + 19.1 -> 19.6 + 17.3 us.  Four moments instead of five (-20 %
// of the filter arithmetic and of the LDS: 7 workgroups per CU instead of 5), 16-byte loads for interior tiles, one
// division, the L1 sum taken where the pixels already are; backward: maps and filtered rows in one buffer (7 workgroups
// per CU instead of 4), conflict-free pitch.  Ablations (no stores -2.4, no horizontal taps -4.3, no vertical taps -1.8,
// no loads -3.4 us) and tile heights 16 / 64 (19.4 / 21.6 us) say no single resource bounds what is left.
#include "skgs_common.h"

namespace skgs {
namespace {

#ifndef SKGS_LOSS_TH
#define SKGS_LOSS_TH 32
#define SKGS_LOSS_VSEG 4
#endif
constexpr int TW   = 32, TH = SKGS_LOSS_TH;      // output tile of one workgroup
constexpr int HALO = 5;                // window radius
constexpr int IW   = TW + 2 * HALO;    // 42 input columns under a tile
constexpr int IH   = TH + 2 * HALO;    // 42 input rows under a tile
constexpr int HP   = TW + 1;           // LDS pitch of the horizontally filtered rows
constexpr int SEG  = 8;                // horizontal pass: outputs per thread (18 inputs -> 8 outputs)
constexpr int VSEG = SKGS_LOSS_VSEG;   // vertical pass: outputs per thread (VSEG + 10 inputs -> VSEG outputs)
constexpr int NT   = TW * TH / VSEG;   // threads of a workgroup: one per VSEG outputs of a column
constexpr int HTASKS = IH * (TW / SEG);  // (row, segment) tasks of the horizontal pass
constexpr int NMOM = 4;                // window moments of the forward: E[x], E[y], E[xx + yy], E[xy]
static_assert(TW % SEG == 0 && TH % VSEG == 0 && NT % 64 == 0 && (NT & (NT - 1)) == 0 && SEG == 8, "tile / thread mapping");
// The BACKWARD walks tiles of its own height: 800 / 32 = 25 tile rows x 25 x 3 channels = 1875 workgroups for 7 x 256 = 1792
// resident ones (66 registers, 21.7 KB of LDS: seven per CU) -- 83 workgroups ran a second round behind all the others.  40 rows per
// tile (five outputs per thread, the same 256 threads) are 1500 workgroups of 25.8 KB, six per CU = 1536: ONE round.  Measured, 8
// alternating bench runs each: image_loss_backward 19.8 -> 17.6 us, the step 0.3431 -> 0.3417 ms; the forward (69 registers, longer
// per-tile chain) gained nothing from the same change and keeps 32.
#ifndef SKGS_LOSS_TH_B
#define SKGS_LOSS_TH_B 40
#define SKGS_LOSS_VSEG_B 5
#endif
constexpr int TH_B = SKGS_LOSS_TH_B, VSEG_B = SKGS_LOSS_VSEG_B, IH_B = TH_B + 2 * HALO, HTASKS_B = IH_B * (TW / SEG);
static_assert(TH_B % VSEG_B == 0 && TW * TH_B / VSEG_B == NT, "the backward's tiles: the same workgroup size");
struct Win {
  float g[11];
};

// Register-blocked separable filter: a thread of the horizontal pass slides the 11-tap window over 18 inputs for 8
// adjacent outputs, a thread of the vertical pass over 14 filtered rows for 4 outputs (3.5 LDS reads per output and
// moment instead of 11).  The inputs go from global memory straight into the registers of the horizontal pass (the
// overlap between neighbouring segments and tiles is served by L1/L2): only the filtered rows live in LDS.
// Interior tiles of images whose rows are 16-byte aligned fetch the 18 inputs [q - 5, q + 13) of the segment at q as
// aligned 16-byte loads of columns [q - 8, q + 16) (6 load instructions per plane instead of 18; the compiler narrows
// the two outer ones to the elements used); everything else takes the zero-padding path.
template <int NMAP, bool INTERIOR>
__device__ __forceinline__ void load_row18(const float* const (&plane)[NMAP], int W, int H, int gy, int gx0,
    float (&v)[NMAP][SEG + 10]) {
  if constexpr (INTERIOR) {
#pragma unroll
    for (int m = 0; m < NMAP; ++m) {
      const float4* row = reinterpret_cast<const float4*>(plane[m] + (size_t) gy * W + (gx0 - 3));
      float4 t[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i] = row[i];
#pragma unroll
      for (int i = 0; i < SEG + 10; ++i) {
        const float4 q = t[(i + 3) >> 2];
        v[m][i] = ((i + 3) & 3) == 0 ? q.x : ((i + 3) & 3) == 1 ? q.y : ((i + 3) & 3) == 2 ? q.z : q.w;
      }
    }
  } else {
    const bool row_ok = gy >= 0 && gy < H;
    const size_t ro   = (size_t) (row_ok ? gy : 0) * W;
#pragma unroll
    for (int i = 0; i < SEG + 10; ++i) {
      const int gx  = gx0 + i;
      const bool ok = row_ok && gx >= 0 && gx < W;
      const size_t o = ro + (ok ? gx : 0);
#pragma unroll
      for (int m = 0; m < NMAP; ++m) v[m][i] = ok ? plane[m][o] : 0.f;
    }
  }
}
// the aligned 24-column windows of a tile's segments start 8 columns left of it and end 8 right of it
__device__ __forceinline__ bool tile_is_interior(int x0, int y0, int W, int H, bool rows_aligned) {
  return rows_aligned && x0 >= 8 && y0 >= HALO && x0 + TW + 8 <= W && y0 + TH + HALO <= H;
}
template <int NMAP>
__device__ __forceinline__ bool planes_aligned(const float* const (&plane)[NMAP], int W) {
  bool ok = (W & 3) == 0;
#pragma unroll
  for (int m = 0; m < NMAP; ++m) ok = ok && (reinterpret_cast<uintptr_t>(plane[m]) & 15) == 0;
  return ok;
}

// Workgroup -> tile.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2; a tile
// shares its 5-pixel halo with its neighbours, so every XCD gets ONE contiguous run of the row-major (channel, tile row,
// tile column) order (~9 tile rows at 800x800): neighbours meet in the same L2 instead of each XCD fetching its own copy
// of every halo from the fabric.
struct TileId {
  int tx, ty, c, linear;
};
__device__ __forceinline__ bool tile_of_block(int tiles_x, int tiles_y, int C, TileId& t) {
  const int n = tiles_x * tiles_y * C, chunk = (n + 7) >> 3;
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int w = xcd * chunk + i;
  if (i >= chunk || w >= n) return false;
  t.linear = w;
  t.tx     = w % tiles_x;
  t.ty     = (w / tiles_x) % tiles_y;
  t.c      = w / (tiles_x * tiles_y);
  return true;
}
inline dim3 tile_grid(int C, int H, int W, int th = TH) {
  const int n = ((W + TW - 1) / TW) * ((H + th - 1) / th) * C;
  return dim3((unsigned) (((n + 7) / 8) * 8));
}

// horizontal pass of the forward: the five moments of 8 adjacent pixels per (row, segment) task into s_h; returns the
// thread's share of sum |x - y| over the tile's own pixels
template <bool INTERIOR>
__device__ __forceinline__ float forward_rows(const float* const (&planes)[2], int W, int H, int x0, int y0, const Win& win,
    float (&s_h)[NMOM][IH][HP]) {
  float l1_sum = 0.f;
  for (int task = threadIdx.x; task < HTASKS; task += NT) {
    const int r = task / (TW / SEG), q0 = (task % (TW / SEG)) * SEG;
    float in[2][SEG + 10];
    load_row18<2, INTERIOR>(planes, W, H, y0 + r - HALO, x0 + q0 - HALO, in);
    if (r >= HALO && r < HALO + TH) {  // (pixels outside the image were loaded as 0 and 0)
#pragma unroll
      for (int i = HALO; i < HALO + SEG; ++i) l1_sum += fabsf(in[0][i] - in[1][i]);
    }
    float a[SEG][NMOM];
#pragma unroll
    for (int o = 0; o < SEG; ++o)
#pragma unroll
      for (int m = 0; m < NMOM; ++m) a[o][m] = 0.f;
#pragma unroll
    for (int i = 0; i < SEG + 10; ++i) {
      const float xv = in[0][i], yv = in[1][i];
      const float sq = xv * xv + yv * yv, xy = xv * yv;
#pragma unroll
      for (int o = 0; o < SEG; ++o) {
        const int k = i - o;  // compile-time after unrolling
        if (k >= 0 && k < 11) {
          const float w = win.g[k];
          a[o][0] += w * xv, a[o][1] += w * yv, a[o][2] += w * sq, a[o][3] += w * xy;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < SEG; ++o)
#pragma unroll
      for (int m = 0; m < NMOM; ++m) s_h[m][r][q0 + o] = a[o][m];
  }
  return l1_sum;
}

__global__ void __launch_bounds__(NT) image_loss_forward_kernel(int C, int H, int W, const float* __restrict__ pred,
    const float* __restrict__ gt, const int32_t* __restrict__ gt_index, Win win, float* __restrict__ dmaps /*[3][C][H][W]*/,
    float* __restrict__ partials) {
  __shared__ float s_h[NMOM][IH][HP];
  __shared__ float s_red[2][NT / 64];
  if (gt_index) gt += (size_t) gt_index[0] * C * H * W;  // image of a [views, C, H, W] stack, chosen on the device
  TileId tile;
  if (!tile_of_block((W + TW - 1) / TW, (H + TH - 1) / TH, C, tile)) return;
  const int c  = tile.c;
  const int x0 = tile.tx * TW, y0 = tile.ty * TH;
  const int tid = threadIdx.x;
  const size_t plane = (size_t) H * W;
  const float* const planes[2] = {pred + c * plane, gt + c * plane};
  const bool interior = tile_is_interior(x0, y0, W, H, planes_aligned<2>(planes, W));
  // |x - y| over the tile's own pixels: summed by the threads that hold them for the horizontal pass
  float l1_sum = interior ? forward_rows<true>(planes, W, H, x0, y0, win, s_h) : forward_rows<false>(planes, W, H, x0, y0, win, s_h);
  __syncthreads();
  const int tx = tid % TW, ty0 = (tid / TW) * VSEG;
  float v[VSEG][NMOM];
#pragma unroll
  for (int o = 0; o < VSEG; ++o)
#pragma unroll
    for (int m = 0; m < NMOM; ++m) v[o][m] = 0.f;
#pragma unroll
  for (int i = 0; i < VSEG + 10; ++i) {
    float hv[NMOM];
#pragma unroll
    for (int m = 0; m < NMOM; ++m) hv[m] = s_h[m][ty0 + i][tx];
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      const int k = i - o;
      if (k >= 0 && k < 11) {
        const float w = win.g[k];
#pragma unroll
        for (int m = 0; m < NMOM; ++m) v[o][m] += w * hv[m];
      }
    }
  }
  float ssim_sum = 0.f;
  const int gx = x0 + tx;
#pragma unroll
  for (int o = 0; o < VSEG; ++o) {
    const int gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      // SSIM needs the two variances only as their sum: one moment E[xx + yy] instead of two
      const float mu1 = v[o][0], mu2 = v[o][1], esq = v[o][2], exy = v[o][3];
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float mu_sq = mu1 * mu1 + mu2 * mu2, mu12 = mu1 * mu2;
      const float s12 = exy - mu12;
      const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu_sq + C1, B2 = (esq - mu_sq) + C2;
      const float inv  = 1.f / (B1 * B2);  // the one division: 1 / B2 = B1 * inv
      const float ssim = A1 * A2 * inv;
      // partial derivatives of ssim w.r.t. the window moments of x (mu1, E[xx], E[xy]) as independent variables
      const float d_mu1 = (2.f * mu2 * (A2 - A1) * inv) - ssim * (2.f * mu1 * (B2 - B1)) * inv;
      const float d_exx = -ssim * (B1 * inv);
      const float d_exy = 2.f * A1 * inv;
      const size_t oo  = (size_t) c * plane + (size_t) gy * W + gx;
      const size_t CHW = (size_t) C * plane;
      stream_store<NT_LOSS_FWD>(dmaps + oo, d_mu1), stream_store<NT_LOSS_FWD>(dmaps + CHW + oo, d_exx);
      stream_store<NT_LOSS_FWD>(dmaps + 2 * CHW + oo, d_exy);
      ssim_sum += ssim;
    }
  }
  // both tile sums in one pass: lanes, then the four waves in a fixed order
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) ssim_sum += __shfl_xor(ssim_sum, d), l1_sum += __shfl_xor(l1_sum, d);
  if ((tid & 63) == 0) s_red[0][tid >> 6] = ssim_sum, s_red[1][tid >> 6] = l1_sum;
  __syncthreads();
  if (tid == 0) {
    const int b = tile.linear;
    float ss = s_red[0][0], ls = s_red[1][0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) ss += s_red[0][w], ls += s_red[1][w];
    partials[2 * b] = ss, partials[2 * b + 1] = ls;
  }
}

// fixed-order sum of the per-tile partials (256 threads of one workgroup) -> loss[3] = {total, L1 mean, SSIM mean}
template <int THREADS>
__device__ __forceinline__ void finalize_loss(int nblocks, double inv_n, float lambda_l1, float lambda_ssim,
    const float* __restrict__ partials, float* __restrict__ loss, double* s_a, double* s_b) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += THREADS) a += partials[2 * i], b += partials[2 * i + 1];
  s_a[threadIdx.x] = a, s_b[threadIdx.x] = b;
  __syncthreads();
  for (int d = THREADS / 2; d > 0; d >>= 1) {
    if (threadIdx.x < d) s_a[threadIdx.x] += s_a[threadIdx.x + d], s_b[threadIdx.x] += s_b[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float ssim_mean = (float) (s_a[0] * inv_n), l1_mean = (float) (s_b[0] * inv_n);
    loss[0] = lambda_l1 * l1_mean + lambda_ssim * (1.0f - ssim_mean);
    loss[1] = l1_mean;
    loss[2] = ssim_mean;
  }
}

__global__ void __launch_bounds__(256) image_loss_finalize_kernel(int nblocks, double inv_n, float lambda_l1,
    float lambda_ssim, const float* __restrict__ partials, float* __restrict__ loss /*[3]*/) {
  __shared__ double s_a[256], s_b[256];
  finalize_loss<256>(nblocks, inv_n, lambda_l1, lambda_ssim, partials, loss, s_a, s_b);
}

__global__ void __launch_bounds__(NT) image_loss_backward_kernel(int C, int H, int W, const float* __restrict__ pred,
    const float* __restrict__ gt, const int32_t* __restrict__ gt_index, Win win, const float* __restrict__ dmaps,
    const float* __restrict__ grad_loss, const float* __restrict__ grad_l1, const float* __restrict__ grad_ssim, float scale_l1,
    float scale_ssim, float* __restrict__ dL_dpred,
    const float* __restrict__ partials, int nblocks, double inv_n, float lambda_l1, float lambda_ssim,
    float* __restrict__ loss3) {
  // the staged maps and, once every thread holds its filtered segment in registers, the filtered rows share one buffer
  // (21.7 KB: 7 workgroups per CU)
  static_assert(HTASKS_B <= NT, "one (row, segment) task per thread");
  constexpr int IP = IW + 1;  // 43: the 8 rows x 4 segments of a 32-lane group read 32 different banks
  __shared__ float s_buf[3 * IH_B * IP];
  float (&s_m)[3][IH_B][IP] = *reinterpret_cast<float (*)[3][IH_B][IP]>(s_buf);
  float (&s_h)[3][IH_B][HP] = *reinterpret_cast<float (*)[3][IH_B][HP]>(s_buf);
  if (gt_index) gt += (size_t) gt_index[0] * C * H * W;
  TileId tile;
  if (!tile_of_block((W + TW - 1) / TW, (H + TH_B - 1) / TH_B, C, tile)) return;
  const int c  = tile.c;
  const int x0 = tile.tx * TW, y0 = tile.ty * TH_B;
  const int tid = threadIdx.x;
  const size_t plane = (size_t) H * W, CHW = (size_t) C * plane;
  const float* const maps[3] = {dmaps + c * plane, dmaps + CHW + c * plane, dmaps + 2 * CHW + c * plane};
  // the own pixels of the epilogue: requested before the filter passes, used after them
  const int tx = tid % TW, ty0 = (tid / TW) * VSEG_B;
  const int gx = x0 + tx;
  float xs[VSEG_B], ys[VSEG_B];
#pragma unroll
  for (int o = 0; o < VSEG_B; ++o) {
    const int gy  = y0 + ty0 + o;
    const bool ok = gx < W && gy < H;
    const size_t oo = (size_t) c * plane + (ok ? (size_t) gy * W + gx : 0);
    xs[o] = pred[oo], ys[o] = gt[oo];
  }
  // the three derivative maps under the tile, staged once (coalesced rows), then the horizontal pass out of LDS
  for (int i = tid; i < IH_B * IW; i += NT) {
    const int r = i / IW, q = i - r * IW;
    const int gy = y0 + r - HALO, gxx = x0 + q - HALO;
    const bool in = gy >= 0 && gy < H && gxx >= 0 && gxx < W;
    const size_t o = in ? (size_t) gy * W + gxx : 0;
    const float m0 = maps[0][o], m1 = maps[1][o], m2 = maps[2][o];
    s_m[0][r][q] = in ? m0 : 0.f, s_m[1][r][q] = in ? m1 : 0.f, s_m[2][r][q] = in ? m2 : 0.f;
  }
  __syncthreads();
  {
    const bool has = tid < HTASKS_B;
    const int r = tid / (TW / SEG), q0 = (tid % (TW / SEG)) * SEG;
    float a[SEG][3];
#pragma unroll
    for (int o = 0; o < SEG; ++o) a[o][0] = a[o][1] = a[o][2] = 0.f;
    if (has) {
#pragma unroll
      for (int i = 0; i < SEG + 10; ++i) {
        const float m0 = s_m[0][r][q0 + i], m1 = s_m[1][r][q0 + i], m2 = s_m[2][r][q0 + i];
#pragma unroll
        for (int o = 0; o < SEG; ++o) {
          const int k = i - o;
          if (k >= 0 && k < 11) {
            const float w = win.g[k];
            a[o][0] += w * m0, a[o][1] += w * m1, a[o][2] += w * m2;
          }
        }
      }
    }
    __syncthreads();  // the filtered rows go where the staged maps were
    if (has) {
#pragma unroll
      for (int o = 0; o < SEG; ++o) s_h[0][r][q0 + o] = a[o][0], s_h[1][r][q0 + o] = a[o][1], s_h[2][r][q0 + o] = a[o][2];
    }
  }
  __syncthreads();
  float v[VSEG_B][3];
#pragma unroll
  for (int o = 0; o < VSEG_B; ++o) v[o][0] = v[o][1] = v[o][2] = 0.f;
#pragma unroll
  for (int i = 0; i < VSEG_B + 10; ++i) {
    const float h0 = s_h[0][ty0 + i][tx], h1 = s_h[1][ty0 + i][tx], h2 = s_h[2][ty0 + i][tx];
#pragma unroll
    for (int o = 0; o < VSEG_B; ++o) {
      const int k = i - o;
      if (k >= 0 && k < 11) {
        const float w = win.g[k];
        v[o][0] += w * h0, v[o][1] += w * h1, v[o][2] += w * h2;
      }
    }
  }
  const float g = grad_loss ? grad_loss[0] : 1.0f;
  // separate cotangents of the two terms (skgs_image_loss_backward_terms: the L1 mean and 1 - SSIM mean are two autograd outputs,
  // each weighted by its own factor outside: networks/losses/build.py:55-64); a missing one is zero.  Wave-uniform branch: the
  // one-cotangent form keeps its expression (and its bits).
  const bool terms = grad_l1 != nullptr || grad_ssim != nullptr;
  const float gs = terms ? (grad_ssim ? grad_ssim[0] : 0.f) * scale_ssim : scale_ssim;
  const float gl = terms ? (grad_l1 ? grad_l1[0] : 0.f) * scale_l1 : scale_l1;
#pragma unroll
  for (int o = 0; o < VSEG_B; ++o) {
    const int gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      const size_t oo = (size_t) c * plane + (size_t) gy * W + gx;
      const float x = xs[o], y = ys[o];
      const float d = x - y;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      const float r = gs * (v[o][0] + 2.f * x * v[o][1] + y * v[o][2]) + gl * sgn;
      stream_store<NT_LOSS_BWD>(dL_dpred + oo, terms ? r : g * r);
    }
  }
  // the loss value itself, when the forward left it to this launch (one workgroup, off everybody else's critical path)
  if (loss3 && tile.linear == 0) {
    __syncthreads();
    double* s_fin = reinterpret_cast<double*>(&s_h[0][0][0]);  // 2 x 256 doubles fit the filtered rows' first plane
    finalize_loss<NT>(nblocks, inv_n, lambda_l1, lambda_ssim, partials, loss3, s_fin, s_fin + NT);
  }
}

Win make_window() {
  // torch: gauss = Tensor([exp(-(x - 5)^2 / (2 sigma^2))]) (fp32), gauss / gauss.sum()   (ssim.py:8-10)
  Win w;
  float s = 0.f;
  for (int i = 0; i < 11; ++i) {
    w.g[i] = (float) exp(-(double) ((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    s += w.g[i];
  }
  for (int i = 0; i < 11; ++i) w.g[i] = w.g[i] / s;
  return w;
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_image_loss_workspace_bytes(int32_t C, int32_t H, int32_t W) {
  const size_t tiles = (size_t) ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * C;
  return (size_t) 3 * C * H * W * 4 + align256(tiles * 2 * 4) + 256;
}

int skgs_image_loss_forward(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt, const int32_t* gt_index,
    float lambda_l1, float lambda_ssim, float* loss3, void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  SKGS_REQUIRE(C > 0 && H > 0 && W > 0 && pred && gt && workspace, "image_loss_forward: bad argument");
  SKGS_REQUIRE(workspace_bytes >= skgs_image_loss_workspace_bytes(C, H, W), "image_loss_forward: workspace too small");
  hipStream_t s   = (hipStream_t) stream;
  float* dmaps    = reinterpret_cast<float*>(workspace);
  float* partials = dmaps + (size_t) 3 * C * H * W;
  ProfScope prof(K_LOSS_FWD, s);
  hipLaunchKernelGGL(image_loss_forward_kernel, tile_grid(C, H, W), dim3(NT), 0, s, C, H, W, pred, gt, gt_index, make_window(), dmaps,
      partials);
  SKGS_CHECK_HIP(hipGetLastError());
  if (loss3) {  // NULL: the caller asks skgs_image_loss_backward for the value (saves this launch)
    const int nblocks = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * C;
    hipLaunchKernelGGL(image_loss_finalize_kernel, dim3(1), dim3(256), 0, s, nblocks, 1.0 / ((double) C * H * W), lambda_l1,
        lambda_ssim, partials, loss3);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

int skgs_image_loss_backward(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt, const int32_t* gt_index,
    float lambda_l1, float lambda_ssim, const float* grad_loss /*device scalar or NULL (=1)*/, const void* workspace,
    size_t workspace_bytes, float* dL_dpred, float* loss3 /*NULL, or where to put the forward's loss values*/,
    skgs_stream_t stream) {
  SKGS_REQUIRE(C > 0 && H > 0 && W > 0 && pred && gt && workspace && dL_dpred, "image_loss_backward: bad argument");
  SKGS_REQUIRE(workspace_bytes >= skgs_image_loss_workspace_bytes(C, H, W), "image_loss_backward: workspace too small");
  const float* dmaps = reinterpret_cast<const float*>(workspace);
  const float n      = (float) ((double) C * H * W);
  const float* partials = dmaps + (size_t) 3 * C * H * W;
  const int nblocks     = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * C;
  ProfScope prof(K_LOSS_BWD, (hipStream_t) stream);
  hipLaunchKernelGGL(image_loss_backward_kernel, tile_grid(C, H, W, TH_B), dim3(NT), 0, (hipStream_t) stream, C, H, W, pred, gt,
      gt_index, make_window(), dmaps, grad_loss, (const float*) nullptr, (const float*) nullptr, lambda_l1 / n, -lambda_ssim / n,
      dL_dpred, partials, nblocks, 1.0 / ((double) C * H * W), lambda_l1, lambda_ssim, loss3);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_image_loss_backward_terms(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt, const int32_t* gt_index,
    const float* grad_l1 /*device scalar or NULL (= 0)*/, const float* grad_ssim /*device scalar or NULL (= 0)*/,
    const void* workspace, size_t workspace_bytes, float* dL_dpred, skgs_stream_t stream) {
  SKGS_REQUIRE(C > 0 && H > 0 && W > 0 && pred && gt && workspace && dL_dpred, "image_loss_backward_terms: bad argument");
  SKGS_REQUIRE(grad_l1 || grad_ssim, "image_loss_backward_terms: at least one cotangent");
  SKGS_REQUIRE(workspace_bytes >= skgs_image_loss_workspace_bytes(C, H, W), "image_loss_backward_terms: workspace too small");
  const float* dmaps = reinterpret_cast<const float*>(workspace);
  const float n      = (float) ((double) C * H * W);
  const float* partials = dmaps + (size_t) 3 * C * H * W;
  const int nblocks     = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * C;
  ProfScope prof(K_LOSS_BWD, (hipStream_t) stream);
  hipLaunchKernelGGL(image_loss_backward_kernel, tile_grid(C, H, W, TH_B), dim3(NT), 0, (hipStream_t) stream, C, H, W, pred, gt,
      gt_index, make_window(), dmaps, (const float*) nullptr, grad_l1, grad_ssim, 1.0f / n, -1.0f / n, dL_dpred, partials, nblocks,
      1.0 / ((double) C * H * W), 1.0f, 1.0f, (float*) nullptr);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
