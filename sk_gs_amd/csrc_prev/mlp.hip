// mlp.hip -- the bone-transform producer of the skeleton stage (scope row (f)-3): frequency encoding + small fused
// linear layers for a FEW rows (one row per bone: B = M ~ 20).
//
// Reference: SimpleDeformationNetwork (networks/sk_gs.py:134-164) = FreqEncoder(joints, degree 10) | FreqEncoder(t,
// degree 6) -> MLP_with_skips(in 76, 8 x 256, skip after layer 4, heads 4 | 4 | 3) (my_ext/blocks/mlp.py:43-85,
// my_ext/_C/src/nerf/freqencoder.cu:7-60).  torch runs it as ~20 launches forward (addmm, relu, cat) and ~45 backward
// for 20 rows -- pure launch latency.  Here one launch per linear layer and direction:
//   forward : Y = act(X1 W[:, :in1]^T + X2 W[:, in1:]^T + b)        (X2 = the skip input, optional)
//   backward: gZ = gY * (Y > 0);  gW = gZ^T [X1 | X2];  gb = sum_b gZ;  gX1 = gZ W[:, :in1], gX2 = gZ W[:, in1:]
//             (each optional, written or added to)
// Workgroup w owns 16 output features (forward, gW, gb) and 16 input features (gX); the whole [B, *] operands are
// staged in LDS (B <= 64 rows per pass).  fp32, products accumulated in the k order of a plain dot product.
#include <algorithm>

#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int LIN_THREADS = 256;
constexpr int LIN_SLAB    = 16;  // features per workgroup
constexpr int LIN_ROWS    = 32;  // rows staged per pass

// out[b, c]: c < D -> x[b, c]; else col = c / D - 1, d = c % D: sin(x[b, d] * 2^(col / 2) + (col % 2) * pi / 2)
__global__ void __launch_bounds__(256) freq_encode_forward_kernel(int B, int D, int deg, const float* __restrict__ x, int ldx,
    float* __restrict__ out, int ldo) {
  const int C = D + 2 * D * deg;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * C) return;
  const int b = t / C, c = t - b * C;
  float v;
  if (c < D) {
    v = x[b * ldx + c];
  } else {
    const int col = c / D - 1, d = c % D;
    v = sinf(scalbnf(x[b * ldx + d], col / 2) + (float) (col % 2) * (3.141592653589793f / 2));
  }
  out[(size_t) b * ldo + c] = v;
}

// grad_x[b, d] = g[b, d] + sum_f 2^f (g[b, D + 2 f D + d] * out[.. + D + d] - g[.. + D + d] * out[..])   (cos = next slot)
__global__ void __launch_bounds__(256) freq_encode_backward_kernel(int B, int D, int deg, const float* __restrict__ g,
    const float* __restrict__ out, int ldo, float* __restrict__ gx, int accumulate) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * D) return;
  const int b = t / D, d = t - b * D;
  const float* gr = g + (size_t) b * ldo;
  const float* o  = out + (size_t) b * ldo;
  float r = gr[d];
  for (int f = 0; f < deg; ++f) {
    const int s = D + 2 * f * D;
    r += scalbnf(1.0f, f) * (gr[s + d] * o[s + D + d] - gr[s + D + d] * o[s + d]);
  }
  gx[t] = accumulate ? gx[t] + r : r;
}

// Copy nrows x ncols floats (row stride ld) into LDS rows of pitch `pitch` starting at column col0, four 16-byte loads
// per thread in flight at a time (a load -> LDS-store loop serialises on the global-load latency: 36 dependent round
// trips made one layer cost 30 us).  ncols % 4 == 0 and 16-byte aligned rows take the vector path.
__device__ __forceinline__ void stage_block(float* s_dst, int pitch, int col0, const float* __restrict__ src, int ld, int nrows,
    int ncols) {
  const bool vec = (ncols & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  if (vec) {
    const int c4 = ncols >> 2, total = nrows * c4;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * LIN_THREADS) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * LIN_THREADS;
        if (i < total) {
          const int r = i / c4, c = i - r * c4;
          v[u] = *reinterpret_cast<const float4*>(src + (size_t) r * ld + 4 * c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * LIN_THREADS;
        if (i < total) {
          const int r = i / c4, c = i - r * c4;
          float* d = s_dst + r * pitch + col0 + 4 * c;
          d[0] = v[u].x, d[1] = v[u].y, d[2] = v[u].z, d[3] = v[u].w;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < nrows * ncols; i += LIN_THREADS) {
      const int r = i / ncols, c = i - r * ncols;
      s_dst[r * pitch + col0 + c] = src[(size_t) r * ld + c];
    }
  }
}

// LDS rows have pitch K + 4 floats (K % 4 == 0 on the fast path): 16-byte aligned for ds_read_b128 and shifted by 4
// banks per row, so 16 lanes reading 16 different rows hit 64 different banks.
__device__ __forceinline__ int lds_pitch(int K) { return (K & 3) == 0 ? K + 4 : (K | 1); }

__global__ void __launch_bounds__(LIN_THREADS) linear_forward_kernel(int B, int in1, int in2, int out, const float* __restrict__ X1,
    int ldx1, const float* __restrict__ X2, int ldx2, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int ldy, int relu) {
  extern __shared__ float s_lin[];
  const int K   = in1 + in2;
  const int Kp  = lds_pitch(K);
  float* s_x    = s_lin;                           // [LIN_ROWS][Kp]
  float* s_w    = s_lin + (size_t) LIN_ROWS * Kp;  // [16][Kp]
  const int o0  = blockIdx.x * LIN_SLAB;
  const int no  = min(LIN_SLAB, out - o0);
  stage_block(s_w, Kp, 0, W + (size_t) o0 * K, K, no, K);
  for (int b0 = 0; b0 < B; b0 += LIN_ROWS) {
    const int nb = min(LIN_ROWS, B - b0);
    __syncthreads();
    stage_block(s_x, Kp, 0, X1 + (size_t) b0 * ldx1, ldx1, nb, in1);
    if (in2) stage_block(s_x, Kp, in1, X2 + (size_t) b0 * ldx2, ldx2, nb, in2);
    __syncthreads();
    // thread -> (output o, rows b = bs, bs + 16)
    const int o = threadIdx.x / 16, bs = threadIdx.x % 16;
    if (o < no) {
      const float bv = bias ? bias[o0 + o] : 0.f;
      const float* wr = s_w + o * Kp;
      for (int b = bs; b < nb; b += 16) {
        const float* xr = s_x + b * Kp;
        float acc = 0.f;
        if ((K & 3) == 0) {
          // the products are added in k order, four at a time from one 16-byte LDS read per operand
#pragma unroll 4
          for (int k = 0; k < K; k += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + k), wv = *reinterpret_cast<const float4*>(wr + k);
            acc += xv.x * wv.x;
            acc += xv.y * wv.y;
            acc += xv.z * wv.z;
            acc += xv.w * wv.w;
          }
        } else {
          for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
        }
        acc += bv;
        Y[(size_t) (b0 + b) * ldy + o0 + o] = relu ? fmaxf(acc, 0.f) : acc;
      }
    }
  }
}

// One launch does both halves: workgroup w computes gW / gb of output features [16w, 16w+16) and gX1 (gX2) of input
// features [16w, 16w+16); the grid covers max(out, in1 + in2) / 16.
__global__ void __launch_bounds__(LIN_THREADS) linear_backward_kernel(int B, int in1, int in2, int out, const float* __restrict__ X1,
    int ldx1, const float* __restrict__ X2, int ldx2, const float* __restrict__ W, const float* __restrict__ Y,
    const float* __restrict__ gY, int ldy, int relu, float* __restrict__ gW, float* __restrict__ gb,
    float* __restrict__ gX1, int ldg1, float* __restrict__ gX2, int ldg2, int accumulate_gx) {
  extern __shared__ float s_lin[];
  const int K  = in1 + in2;
  const int Kp = lds_pitch(K), Op = lds_pitch(out);
  float* s_x   = s_lin;                            // [B][Kp]
  float* s_g   = s_x + (size_t) B * Kp;            // [B][Op]    gZ = gY * relu'
  float* s_wc  = s_g + (size_t) B * Op;            // [out][20]  W[:, k0 : k0 + 16]
  const int k0 = blockIdx.x * LIN_SLAB;
  const bool do_gx = k0 < K && (gX1 || gX2);
  const int nk = do_gx ? min(LIN_SLAB, K - k0) : 0;
  stage_block(s_x, Kp, 0, X1, ldx1, B, in1);
  if (in2) stage_block(s_x, Kp, in1, X2, ldx2, B, in2);
  stage_block(s_g, Op, 0, gY, ldy, B, out);
  if (do_gx) stage_block(s_wc, LIN_SLAB + 4, 0, W + k0, K, out, nk);
  if (relu) {  // gZ = gY * (Y > 0)
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < B * out; i0 += 4 * LIN_THREADS) {
      float y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * LIN_THREADS;
        if (i < B * out) y[u] = Y[(size_t) (i / out) * ldy + (i % out)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * LIN_THREADS;
        if (i < B * out && !(y[u] > 0.f)) s_g[(i / out) * Op + (i % out)] = 0.f;
      }
    }
  }
  __syncthreads();
  // ---- gW, gb of this workgroup's output slab: gW[o, k] = sum_b gZ[b, o] X[b, k]
  const int o0 = blockIdx.x * LIN_SLAB;
  if (o0 < out) {
    const int no = min(LIN_SLAB, out - o0);
    if ((K & 3) == 0) {
      const int K4 = K >> 2;
      for (int i = threadIdx.x; i < no * K4; i += LIN_THREADS) {
        const int o = i / K4, k = 4 * (i - o * K4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
          const float g   = s_g[b * Op + o0 + o];
          const float4 xv = *reinterpret_cast<const float4*>(s_x + b * Kp + k);
          acc.x += g * xv.x, acc.y += g * xv.y, acc.z += g * xv.z, acc.w += g * xv.w;
        }
        float* d = gW + (size_t) (o0 + o) * K + k;  // (gW may be a 4-byte aligned slice of a flat gradient buffer)
        d[0] = acc.x, d[1] = acc.y, d[2] = acc.z, d[3] = acc.w;
      }
    } else {
      for (int i = threadIdx.x; i < no * K; i += LIN_THREADS) {
        const int o = i / K, k = i - o * K;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += s_g[b * Op + o0 + o] * s_x[b * Kp + k];
        gW[(size_t) (o0 + o) * K + k] = acc;
      }
    }
    if (gb && threadIdx.x < no) {
      float acc = 0.f;
      for (int b = 0; b < B; ++b) acc += s_g[b * Op + o0 + threadIdx.x];
      gb[o0 + threadIdx.x] = acc;
    }
  }
  // ---- gX of this workgroup's input slab: gX[b, k] = sum_o gZ[b, o] W[o, k]
  if (do_gx) {
    for (int i = threadIdx.x; i < B * nk; i += LIN_THREADS) {
      const int b = i / nk, kk = i - b * nk, k = k0 + kk;
      float* dst = k < in1 ? (gX1 ? gX1 + (size_t) b * ldg1 + k : nullptr) : (gX2 ? gX2 + (size_t) b * ldg2 + (k - in1) : nullptr);
      if (!dst) continue;
      float acc = 0.f;
#pragma unroll 8
      for (int o = 0; o < out; ++o) acc += s_g[b * Op + o] * s_wc[o * (LIN_SLAB + 4) + kk];
      if (k >= in1 ? (accumulate_gx & 2) : (accumulate_gx & 1))
        *dst += acc;
      else
        *dst = acc;
    }
  }
}

}  // namespace
}  // namespace skgs

using namespace skgs;

// both linear kernels may stage more than the default 64 KB of dynamic LDS
static int allow_large_lds() {
  static int rc = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_forward_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
          160 * 1024);
    return e == hipSuccess ? 0 : 1;
  }();
  return rc;
}

extern "C" {

int skgs_freq_encode_forward(int32_t B, int32_t D, int32_t degree, const float* x, int32_t ld_x, float* out,
    int32_t ld_out, skgs_stream_t stream) {
  SKGS_REQUIRE(B >= 0 && D >= 1 && degree >= 0, "freq_encode: bad sizes");
  if (B == 0) return 0;
  const int C = D + 2 * D * degree;
  SKGS_REQUIRE(x && out && ld_out >= C, "freq_encode: NULL argument or ld_out < D + 2 D degree");
  SKGS_REQUIRE(ld_x == 0 || ld_x >= D, "freq_encode: ld_x must be 0 (one row for all B) or >= D");
  hipLaunchKernelGGL(freq_encode_forward_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t) stream, B, D, degree, x,
      ld_x, out, ld_out);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_freq_encode_backward(int32_t B, int32_t D, int32_t degree, const float* grad_out, const float* out, int32_t ld_out,
    float* grad_x, int32_t accumulate, skgs_stream_t stream) {
  SKGS_REQUIRE(B >= 0 && D >= 1 && degree >= 0, "freq_encode: bad sizes");
  if (B == 0) return 0;
  SKGS_REQUIRE(grad_out && out && grad_x, "freq_encode_backward: NULL argument");
  hipLaunchKernelGGL(freq_encode_backward_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t) stream, B, D, degree,
      grad_out, out, ld_out, grad_x, accumulate);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_linear_forward(int32_t B, int32_t in1, int32_t in2, int32_t out, const float* X1, int32_t ldx1, const float* X2,
    int32_t ldx2, const float* W, const float* bias, float* Y, int32_t ldy, int32_t relu, skgs_stream_t stream) {
  SKGS_REQUIRE(B >= 0 && in1 >= 1 && in2 >= 0 && out >= 1, "linear_forward: bad sizes");
  if (B == 0) return 0;
  SKGS_REQUIRE(X1 && W && Y && (in2 == 0 || X2), "linear_forward: NULL argument");
  const int K      = in1 + in2;
  const size_t lds = ((size_t) LIN_ROWS + LIN_SLAB) * ((K & 3) == 0 ? K + 4 : (K | 1)) * 4;
  SKGS_REQUIRE(lds <= 160 * 1024, "linear_forward: in1 + in2 too large for the LDS staging");
  SKGS_REQUIRE(lds <= 64 * 1024 || allow_large_lds() == 0, "linear_forward: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(linear_forward_kernel, dim3((out + LIN_SLAB - 1) / LIN_SLAB), dim3(LIN_THREADS), lds, (hipStream_t) stream,
      B, in1, in2, out, X1, ldx1, X2, ldx2, W, bias, Y, ldy, relu);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_linear_backward(int32_t B, int32_t in1, int32_t in2, int32_t out, const float* X1, int32_t ldx1, const float* X2,
    int32_t ldx2, const float* W, const float* Y, const float* gY, int32_t ldy, int32_t relu, float* gW, float* gb,
    float* gX1, int32_t ldg1, float* gX2, int32_t ldg2, int32_t accumulate_gx, skgs_stream_t stream) {
  SKGS_REQUIRE(B >= 1 && in1 >= 1 && in2 >= 0 && out >= 1, "linear_backward: bad sizes");
  SKGS_REQUIRE(X1 && W && gY && gW && (in2 == 0 || X2) && (!relu || Y), "linear_backward: NULL argument");
  const int K      = in1 + in2;
  const size_t lds = ((size_t) B * (((K & 3) == 0 ? K + 4 : (K | 1)) + ((out & 3) == 0 ? out + 4 : (out | 1))) +
                      (size_t) out * (LIN_SLAB + 4)) * 4;
  SKGS_REQUIRE(lds <= 160 * 1024, "linear_backward: B x (in + out) too large for the LDS staging");
  SKGS_REQUIRE(lds <= 64 * 1024 || allow_large_lds() == 0, "linear_backward: cannot raise the dynamic LDS limit");
  const int slabs = (std::max(out, K) + LIN_SLAB - 1) / LIN_SLAB;
  hipLaunchKernelGGL(linear_backward_kernel, dim3(slabs), dim3(LIN_THREADS), lds, (hipStream_t) stream, B, in1, in2, out, X1,
      ldx1, X2, ldx2, W, Y, gY, ldy, relu, gW, gb, gX1, ldg1, gX2, ldg2, accumulate_gx);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
