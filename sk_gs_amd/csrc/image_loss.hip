// image_loss.hip -- fused training-image loss  lambda_l1 * mean|x - y| + lambda_ssim * (1 - mean SSIM(x, y))
// forward and backward (gfx950).  Scope-table row (f)-1: "fused image loss".
//
// Reference: networks/losses/ssim.py:20-62 (11x11 Gaussian window sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2,
// five depth-wise conv2d + ~15 element-wise torch kernels forward, as many again backward), networks/losses/
// image_loss.py:6-32 (L1 mean), weights exps/default.yaml:83-84.  On MI355X the depth-wise convs run in MIOpen at
// ~0.4 ms each -- 3.5 ms per training step, more than the whole rasterizer.
//
// Here: the window is separable, so each 32x32 output tile stages a 42x42 halo of x and y in LDS, runs the 11-tap
// horizontal pass for the 5 moments (x, y, xx, yy, xy) into LDS and the vertical pass into registers (both passes
// register-blocked: a thread slides the window over 18 / 14 inputs for 8 / 4 outputs), evaluates the SSIM map and its
// partial derivatives w.r.t. (mu1, E[xx], E[xy]) in place, and block-reduces the SSIM and L1 sums.
// Backward = the same separable convolution applied to the three derivative maps:
//     dL/dx = gs * (w * dmu1 + 2 x (w * dExx) + y (w * dExy)) + gl * sign(x - y).
// HBM traffic: forward reads 2 and writes 3 image planes, backward reads 5 and writes 1 (vs ~60 plane passes in
// the torch graph).  Per-tile partial sums are reduced in a fixed order: the loss value is bitwise reproducible.
#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int TW   = 32, TH = 32;      // output tile of one workgroup
constexpr int HALO = 5;                // window radius
constexpr int IW   = TW + 2 * HALO;    // 42 staged columns
constexpr int IH   = TH + 2 * HALO;    // 42 staged rows
constexpr int IP   = IW + 2;           // LDS pitch of staged inputs (backward)
constexpr int HP   = TW + 1;           // LDS pitch of the horizontally filtered rows
constexpr int SEG  = 8;                // horizontal pass: outputs per thread (18 inputs -> 8 outputs)
constexpr int VSEG = 4;                // vertical pass: outputs per thread (14 inputs -> 4 outputs)
static_assert(TW % SEG == 0 && (TH / VSEG) * TW == 256 && IH * (TW / SEG) <= 256, "tile / thread mapping");
struct Win {
  float g[11];
};

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wid] = v;
  __syncthreads();
  const float r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  return r;
}

// Register-blocked separable filter: a thread of the horizontal pass slides the 11-tap window over 18 inputs for 8
// adjacent outputs, a thread of the vertical pass over 14 filtered rows for 4 outputs (3.5 LDS reads per output and
// moment instead of 11).  The inputs go from global memory straight into the registers of the horizontal pass (the
// overlap between neighbouring segments and tiles is served by L1/L2): only the filtered rows live in LDS (27 KB per
// workgroup, 5 workgroups per CU).  Interior tiles skip the zero-padding tests.
template <int NMAP>
__device__ __forceinline__ void load_row18(const float* const (&plane)[NMAP], int W, int H, int gy, int gx0, bool interior,
    float (&v)[NMAP][SEG + 10]) {
  if (interior) {
#pragma unroll
    for (int m = 0; m < NMAP; ++m) {
      const float* row = plane[m] + (size_t) gy * W + gx0;
#pragma unroll
      for (int i = 0; i < SEG + 10; ++i) v[m][i] = row[i];
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
inline dim3 tile_grid(int C, int H, int W) {
  const int n = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * C;
  return dim3((unsigned) (((n + 7) / 8) * 8));
}

__global__ void __launch_bounds__(256) image_loss_forward_kernel(int C, int H, int W, const float* __restrict__ pred,
    const float* __restrict__ gt, const int32_t* __restrict__ gt_index, Win win, float* __restrict__ dmaps /*[3][C][H][W]*/,
    float* __restrict__ partials) {
  __shared__ float s_h[5][IH][HP];
  __shared__ float s_red[4];
  if (gt_index) gt += (size_t) gt_index[0] * C * H * W;  // image of a [views, C, H, W] stack, chosen on the device
  TileId tile;
  if (!tile_of_block((W + TW - 1) / TW, (H + TH - 1) / TH, C, tile)) return;
  const int c  = tile.c;
  const int x0 = tile.tx * TW, y0 = tile.ty * TH;
  const int tid = threadIdx.x;
  const size_t plane = (size_t) H * W;
  const float* const planes[2] = {pred + c * plane, gt + c * plane};
  const bool interior = x0 >= HALO && y0 >= HALO && x0 + TW + HALO <= W && y0 + TH + HALO <= H;
  if (tid < IH * (TW / SEG)) {
    const int r = tid / (TW / SEG), q0 = (tid % (TW / SEG)) * SEG;
    float in[2][SEG + 10];
    load_row18<2>(planes, W, H, y0 + r - HALO, x0 + q0 - HALO, interior, in);
    float a[SEG][5];
#pragma unroll
    for (int o = 0; o < SEG; ++o)
#pragma unroll
      for (int m = 0; m < 5; ++m) a[o][m] = 0.f;
#pragma unroll
    for (int i = 0; i < SEG + 10; ++i) {
      const float xv = in[0][i], yv = in[1][i];
      const float xx = xv * xv, yy = yv * yv, xy = xv * yv;
#pragma unroll
      for (int o = 0; o < SEG; ++o) {
        const int k = i - o;  // compile-time after unrolling
        if (k >= 0 && k < 11) {
          const float w = win.g[k];
          a[o][0] += w * xv, a[o][1] += w * yv, a[o][2] += w * xx, a[o][3] += w * yy, a[o][4] += w * xy;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < SEG; ++o)
#pragma unroll
      for (int m = 0; m < 5; ++m) s_h[m][r][q0 + o] = a[o][m];
  }
  __syncthreads();
  const int tx = tid % TW, ty0 = (tid / TW) * VSEG;
  float v[VSEG][5];
#pragma unroll
  for (int o = 0; o < VSEG; ++o)
#pragma unroll
    for (int m = 0; m < 5; ++m) v[o][m] = 0.f;
#pragma unroll
  for (int i = 0; i < VSEG + 10; ++i) {
    float hv[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) hv[m] = s_h[m][ty0 + i][tx];
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      const int k = i - o;
      if (k >= 0 && k < 11) {
        const float w = win.g[k];
#pragma unroll
        for (int m = 0; m < 5; ++m) v[o][m] += w * hv[m];
      }
    }
  }
  float ssim_sum = 0.f, l1_sum = 0.f;
  const int gx = x0 + tx;
#pragma unroll
  for (int o = 0; o < VSEG; ++o) {
    const int gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      const float mu1 = v[o][0], mu2 = v[o][1], exx = v[o][2], eyy = v[o][3], exy = v[o][4];
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
      const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
      const float inv  = 1.f / (B1 * B2);
      const float ssim = A1 * A2 * inv;
      // partial derivatives of ssim w.r.t. the window moments of x (mu1, E[xx], E[xy]) as independent variables
      const float d_mu1 = (2.f * mu2 * (A2 - A1) * inv) - ssim * (2.f * mu1 * (B2 - B1)) * inv;
      const float d_exx = -ssim / B2;
      const float d_exy = 2.f * A1 * inv;
      const size_t po  = (size_t) gy * W + gx;
      const size_t oo  = (size_t) c * plane + po;
      const size_t CHW = (size_t) C * plane;
      dmaps[oo] = d_mu1, dmaps[CHW + oo] = d_exx, dmaps[2 * CHW + oo] = d_exy;
      ssim_sum += ssim;
      l1_sum += fabsf(planes[0][po] - planes[1][po]);
    }
  }
  const float ssum = block_sum_256(ssim_sum, s_red);
  const float lsum = block_sum_256(l1_sum, s_red);
  if (tid == 0) {
    const int b = tile.linear;
    partials[2 * b] = ssum, partials[2 * b + 1] = lsum;
  }
}

// fixed-order sum of the per-tile partials (256 threads of one workgroup) -> loss[3] = {total, L1 mean, SSIM mean}
__device__ __forceinline__ void finalize_loss(int nblocks, double inv_n, float lambda_l1, float lambda_ssim,
    const float* __restrict__ partials, float* __restrict__ loss, double* s_a, double* s_b) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += partials[2 * i], b += partials[2 * i + 1];
  s_a[threadIdx.x] = a, s_b[threadIdx.x] = b;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
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
  finalize_loss(nblocks, inv_n, lambda_l1, lambda_ssim, partials, loss, s_a, s_b);
}

__global__ void __launch_bounds__(256) image_loss_backward_kernel(int C, int H, int W, const float* __restrict__ pred,
    const float* __restrict__ gt, const int32_t* __restrict__ gt_index, Win win, const float* __restrict__ dmaps,
    const float* __restrict__ grad_loss, float scale_l1, float scale_ssim, float* __restrict__ dL_dpred,
    const float* __restrict__ partials, int nblocks, double inv_n, float lambda_l1, float lambda_ssim,
    float* __restrict__ loss3) {
  __shared__ float s_m[3][IH][IP];
  __shared__ float s_h[3][IH][HP];
  if (gt_index) gt += (size_t) gt_index[0] * C * H * W;
  TileId tile;
  if (!tile_of_block((W + TW - 1) / TW, (H + TH - 1) / TH, C, tile)) return;
  const int c  = tile.c;
  const int x0 = tile.tx * TW, y0 = tile.ty * TH;
  const int tid = threadIdx.x;
  const size_t plane = (size_t) H * W, CHW = (size_t) C * plane;
  for (int i = tid; i < IH * IW; i += 256) {
    const int r = i / IW, q = i - r * IW;
    const int gy = y0 + r - HALO, gx = x0 + q - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = (size_t) c * plane + (size_t) gy * W + gx;
    s_m[0][r][q] = in ? dmaps[o] : 0.f;
    s_m[1][r][q] = in ? dmaps[CHW + o] : 0.f;
    s_m[2][r][q] = in ? dmaps[2 * CHW + o] : 0.f;
  }
  __syncthreads();
  if (tid < IH * (TW / SEG)) {
    const int r = tid / (TW / SEG), q0 = (tid % (TW / SEG)) * SEG;
    float a[SEG][3];
#pragma unroll
    for (int o = 0; o < SEG; ++o) a[o][0] = a[o][1] = a[o][2] = 0.f;
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
#pragma unroll
    for (int o = 0; o < SEG; ++o) s_h[0][r][q0 + o] = a[o][0], s_h[1][r][q0 + o] = a[o][1], s_h[2][r][q0 + o] = a[o][2];
  }
  __syncthreads();
  const int tx = tid % TW, ty0 = (tid / TW) * VSEG;
  float v[VSEG][3];
#pragma unroll
  for (int o = 0; o < VSEG; ++o) v[o][0] = v[o][1] = v[o][2] = 0.f;
#pragma unroll
  for (int i = 0; i < VSEG + 10; ++i) {
    const float h0 = s_h[0][ty0 + i][tx], h1 = s_h[1][ty0 + i][tx], h2 = s_h[2][ty0 + i][tx];
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      const int k = i - o;
      if (k >= 0 && k < 11) {
        const float w = win.g[k];
        v[o][0] += w * h0, v[o][1] += w * h1, v[o][2] += w * h2;
      }
    }
  }
  const int gx = x0 + tx;
  const float g = grad_loss ? grad_loss[0] : 1.0f;
#pragma unroll
  for (int o = 0; o < VSEG; ++o) {
    const int gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      const size_t oo = (size_t) c * plane + (size_t) gy * W + gx;
      const float x = pred[oo], y = gt[oo];
      const float d = x - y;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      dL_dpred[oo] = g * (scale_ssim * (v[o][0] + 2.f * x * v[o][1] + y * v[o][2]) + scale_l1 * sgn);
    }
  }
  // the loss value itself, when the forward left it to this launch (one workgroup, off everybody else's critical path)
  if (loss3 && tile.linear == 0) {
    __syncthreads();
    double* s_a = reinterpret_cast<double*>(&s_m[0][0][0]);  // 2 x 256 doubles fit the first staging plane
    finalize_loss(nblocks, inv_n, lambda_l1, lambda_ssim, partials, loss3, s_a, s_a + 256);
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
  hipLaunchKernelGGL(image_loss_forward_kernel, tile_grid(C, H, W), dim3(256), 0, s, C, H, W, pred, gt, gt_index, make_window(), dmaps,
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
  hipLaunchKernelGGL(image_loss_backward_kernel, tile_grid(C, H, W), dim3(256), 0, (hipStream_t) stream, C, H, W, pred, gt,
      gt_index, make_window(), dmaps, grad_loss, lambda_l1 / n, -lambda_ssim / n, dL_dpred, partials, nblocks,
      1.0 / ((double) C * H * W), lambda_l1, lambda_ssim, loss3);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
