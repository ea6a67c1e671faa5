// mfma4x4_layer.hip -- can a 256 x 256 fp32 layer on FEW rows (4 or 8 per workgroup) run at the weight-streaming rate of a CU?
// (csrc/sp_mlp.hip: 512 rows are 32 row blocks of 16 for v_mfma_f32_16x16x4_f32 -- 32 of 256 CUs, MFMA-issue bound at 3.4 us
// per layer.  v_mfma_f32_4x4x1_16B_f32 = 16 independent 4 x 4 outer products per instruction at the same MAC rate: 4-row tiles.)
//
//   forward, weights [n][k] (nn.Linear): blocks <-> k.  One dwordx4 load per lane = a [4 n x 64 k] block as 4 rows x 256 B (full
//   lines); register q of lane (b, j) = W[n0 + j][k0 + 4 b + q] is the B operand of MFMA q, whose A operand is
//   x[i][k0 + 4 b + q] (one ds_read_b128 per k0 step).  16 partial sums per output (one per block) are added at the layer's end.
//
// Checks the operand layout against a host product, then times 8 layers on `nwg` workgroups.
// build: hipcc -O3 --offload-arch=gfx950 mfma4x4_layer.hip -o mfma4x4_layer ; run: ./mfma4x4_layer [nwg]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 256, LAYERS = 8, PITCH = 260;
typedef float v4f __attribute__((ext_vector_type(4)));

template <int RG, bool REDUCE>
__global__ void __launch_bounds__(256) layers_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y,
    int layers) {
  constexpr int ROWS = 4 * RG;
  __shared__ float h[2][ROWS][PITCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = lane >> 2, j = lane & 3;
  const float* Xb = X + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += 256) h[0][t / N][t % N] = Xb[t];
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < layers; ++l) {
    const float* Wl = W + (size_t) (l % LAYERS) * N * N + (size_t) (64 * wave + j) * N + 4 * b;
    v4f acc[RG][16];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[rg][g] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k0 = 0; k0 < N; k0 += 64) {
      float4 wv[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) wv[g] = *reinterpret_cast<const float4*>(Wl + (size_t) (4 * g) * N + k0);
      float4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const float4*>(&h[cur][4 * rg + j][k0 + 4 * b]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          acc[rg][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].x, wv[g].x, acc[rg][g], 0, 0, 0);
          acc[rg][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].y, wv[g].y, acc[rg][g], 0, 0, 0);
          acc[rg][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].z, wv[g].z, acc[rg][g], 0, 0, 0);
          acc[rg][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].w, wv[g].w, acc[rg][g], 0, 0, 0);
        }
    }
    // sum over the 16 blocks (lanes with the same j), then lanes b == 0 hold feature 64 w + 4 g + j of rows 4 rg + i
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[rg][g][i];
          if (REDUCE) {
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
          }
          if (b == 0) h[cur ^ 1][4 * rg + i][64 * wave + 4 * g + j] = layers > 1 ? fmaxf(v, 0.f) * 0.1f : v;
        }
    cur ^= 1;
    __syncthreads();
  }
  float* Yb = Y + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += 256) Yb[t] = h[cur][t / N][t % N];
}


// ---- the same layer chain, software-pipelined: the weights of a WHOLE layer are in flight (4 k0-steps x 16 dwordx4 per lane =
// 256 VGPRs), each step's registers refilled with the next layer's block right after its MFMAs; the 16 block sums as a transposing
// tree (row_ror within a row of 16 under bank masks, then permlane32 / permlane16 swaps): 64 -> 32 -> 16 -> 8 -> 4 registers, after
// which every lane holds 4 finished outputs
#define ROR_ADD(dst, src, n, banks) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:" banks : "+v"(dst) : "v"(src))
__device__ __forceinline__ void tree16(v4f (&acc)[16], float (&out)[4]) {
  float r[64];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) r[4 * g + i] = acc[g][i];
  // stage 1 (lane bit 3): pairs (2p, 2p+1) -> banks {0,1}: r[2p], banks {2,3}: r[2p+1]
#pragma unroll
  for (int p = 0; p < 32; ++p) {
    asm volatile("s_nop 1");
    ROR_ADD(r[2 * p], r[2 * p], 8, "0x3");
    ROR_ADD(r[2 * p], r[2 * p + 1], 8, "0xc");
  }
  // stage 2 (lane bit 2): pairs of stage-1 results (4p, 4p+2) -> banks {0,2}: first, banks {1,3}: second
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    asm volatile("s_nop 1");
    ROR_ADD(r[4 * p], r[4 * p], 4, "0x5");
    ROR_ADD(r[4 * p], r[4 * p + 2], 12, "0xa");
  }
  // stage 3 (lane bit 5) and 4 (lane bit 4): swaps
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r[8 * p]), "+v"(r[8 * p + 4]));
    r[8 * p] += r[8 * p + 4];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r[16 * p]), "+v"(r[16 * p + 8]));
    r[16 * p] += r[16 * p + 8];
    out[p] = r[16 * p];
  }
}

__global__ void __launch_bounds__(256) layers_pipelined_kernel(const float* __restrict__ W, const float* __restrict__ X,
    float* __restrict__ Y, int layers, int decode) {
  constexpr int ROWS = 4;
  __shared__ float h[2][ROWS][PITCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = lane >> 2, j = lane & 3;
  const float* Xb = X + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += 256) h[0][t / N][t % N] = Xb[t];
  float4 wv[4][16];
  const float* W0 = W + (size_t) (64 * wave + j) * N + 4 * b;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int g = 0; g < 16; ++g) wv[s][g] = *reinterpret_cast<const float4*>(W0 + (size_t) (4 * g) * N + 64 * s);
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < layers; ++l) {
    const float* Wn = W0 + (size_t) ((l + 1) % LAYERS) * N * N;
    v4f acc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(&h[cur][j][64 * s + 4 * b]);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, wv[s][g].x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, wv[s][g].y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, wv[s][g].z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, wv[s][g].w, acc[g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; ++g) wv[s][g] = *reinterpret_cast<const float4*>(Wn + (size_t) (4 * g) * N + 64 * s);
      __builtin_amdgcn_sched_barrier(0);
    }
    float out[4];
    tree16(acc, out);
    // decode: after the tree, register p of lane L holds ... (found by the probe below: every lane writes (lane, p, value))
    if (decode) {
#pragma unroll
      for (int p = 0; p < 4; ++p) Y[(size_t) blockIdx.x * 1024 + (wave * 64 + lane) * 4 + p] = out[p];
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) h[cur ^ 1][p][64 * wave + lane] = fmaxf(out[p], 0.f) * 0.1f;  // (timing only: any bijection)
    }
    cur ^= 1;
    __syncthreads();
  }
  if (!decode) {
    float* Yb = Y + (size_t) blockIdx.x * ROWS * N;
    for (int t = threadIdx.x; t < ROWS * N; t += 256) Yb[t] = h[cur][t / N][t % N];
  }
}


// ---- weights [r][out] (the reduction index is the ROW: the backward's direction for nn.Linear storage, the forward's for a
// transposed copy): blocks <-> outputs, no block sums.  Lane l loads outputs 4 l .. 4 l + 3 of row r (one row x 1 KB per wave
// instruction), register q is the B operand of MFMA q (block b, column j <-> output 16 b + 4 j + q); the A operand x[i][r] is the
// same in every block.  The 4 waves split the reduction rows (64 each); their partial tiles meet in LDS.
template <int RG, int NW, int MODE = 0>
__global__ void __launch_bounds__(64 * NW) layers_rowmajor_kernel(const float* __restrict__ W, const float* __restrict__ X,
    float* __restrict__ Y, int layers) {
  constexpr int ROWS = 4 * RG, RPW = N / NW, NT = 64 * NW;  // reduction rows per wave: ALL of a layer's share is in flight
  __shared__ float h[2][ROWS][PITCH];
  __shared__ float part[NW][ROWS][N];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i_ = lane & 3;
  const float* Xb = X + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += NT) h[0][t / N][t % N] = Xb[t];
  const float* W0 = W + (size_t) (RPW * wave) * N + 4 * lane;
  float4 wv[RPW];
#pragma unroll
  for (int d = 0; d < RPW; ++d) wv[d] = *reinterpret_cast<const float4*>(W0 + (size_t) d * N);
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int l = 0; l < LAYERS; ++l) {
    if (l >= layers) break;
    const float* Wn = W0 + (size_t) ((l + 1) % LAYERS) * N * N;
    v4f acc[RG][4];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rg][q] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d4 = 0; d4 < RPW / 4; ++d4) {
      float4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const float4*>(&h[cur][4 * rg + i_][RPW * wave + 4 * d4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = 4 * d4 + e;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          const float av = e == 0 ? a[rg].x : e == 1 ? a[rg].y : e == 2 ? a[rg].z : a[rg].w;
          if (MODE == 1) {
            acc[rg][0][0] += av * wv[d].x, acc[rg][1][0] += av * wv[d].y, acc[rg][2][0] += av * wv[d].z, acc[rg][3][0] += av * wv[d].w;
          } else {
          acc[rg][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].x, acc[rg][0], 0, 0, 0);
          acc[rg][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].y, acc[rg][1], 0, 0, 0);
          acc[rg][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].z, acc[rg][2], 0, 0, 0);
          acc[rg][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].w, acc[rg][3], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        wv[d] = *reinterpret_cast<const float4*>(Wn + (size_t) d * N);  // refill with the next layer's row
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(&part[wave][4 * rg + i][4 * lane]) = make_float4(acc[rg][0][i], acc[rg][1][i], acc[rg][2][i], acc[rg][3][i]);
    __syncthreads();
    for (int t = threadIdx.x; t < ROWS * N / 4; t += NT) {
      const int r = t / (N / 4), c = 4 * (t % (N / 4));
      float4 s0 = *reinterpret_cast<const float4*>(&part[0][r][c]);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(&part[w][r][c]);
        s0.x += p.x, s0.y += p.y, s0.z += p.z, s0.w += p.w;
      }
      if (layers > 1) s0 = make_float4(fmaxf(s0.x, 0.f) * 0.1f, fmaxf(s0.y, 0.f) * 0.1f, fmaxf(s0.z, 0.f) * 0.1f, fmaxf(s0.w, 0.f) * 0.1f);
      *reinterpret_cast<float4*>(&h[cur ^ 1][r][c]) = s0;
    }
    cur ^= 1;
    __syncthreads();
  }
  float* Yb = Y + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += NT) Yb[t] = h[cur][t / N][t % N];
}

template <int RG, int NW, int MODE = 0>
void run_rowmajor(const float* W, const float* X, float* Y, int nwg, int reps, const std::vector<float>& hW, const std::vector<float>& hX) {
  hipLaunchKernelGGL((layers_rowmajor_kernel<RG, NW, MODE>), dim3(1), dim3(64 * NW), 0, 0, W, X, Y, 1);
  std::vector<float> hY(4 * RG * N);
  hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 4 * RG; ++r)
    for (int n = 0; n < N; ++n) {
      double s2 = 0;
      for (int k = 0; k < N; ++k) s2 += (double) hX[r * N + k] * hW[(size_t) k * N + n];
      worst = fmax(worst, fabs(s2 - hY[r * N + n]));
    }
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((layers_rowmajor_kernel<RG, NW, MODE>), dim3(nwg), dim3(64 * NW), 0, 0, W, X, Y, LAYERS);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((layers_rowmajor_kernel<RG, NW, MODE>), dim3(nwg), dim3(64 * NW), 0, 0, W, X, Y, LAYERS);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("  %d rows/WG, weights [r][out], a layer in flight, split-K over %d waves, mode %d: %7.2f  %6.2f   (one layer vs host: %.2g)\n",
         4 * RG, NW, MODE, ms / reps * 1e3f, ms / reps * 1e3f / LAYERS, worst);
}


// ---- weights [r][out] without split-K: wave w owns outputs 64 w .. 64 w + 63 for ALL reduction rows.  One dwordx4 load per lane =
// a [4 r x 64 out] block (lane l: row r0 + l / 16, outputs 64 w + 4 (l % 16) .. + 3: 4 rows x 256 B, 16 consecutive lanes
// contiguous); block b of MFMA q multiplies x[i][r0 + b / 4] by outputs 64 w + 16 (b % 4) + 4 j + q.  The four lane rows hold
// partial sums over r mod 4: two swap stages (permlane32, permlane16) leave each lane with 4 finished outputs.  No LDS partials,
// one barrier per layer; the whole next layer (64 loads per wave) is in flight.
__global__ void __launch_bounds__(256) layers_quarter_kernel(const float* __restrict__ W, const float* __restrict__ X,
    float* __restrict__ Y, int layers, int decode) {
  constexpr int ROWS = 4;
  // activations as h[buffer][row][r mod 4][r / 4]: the A operand of lane (b, i) for steps d .. d + 3 is one ds_read_b128
  __shared__ float h[2][ROWS][4][68];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i_ = lane & 3, b = lane >> 2;
  const float* Xb = X + (size_t) blockIdx.x * ROWS * N;
  for (int t = threadIdx.x; t < ROWS * N; t += 256) h[0][t / N][(t % N) & 3][(t % N) >> 2] = Xb[t];
  const float* W0 = W + (size_t) (lane >> 4) * N + 64 * wave + 4 * (lane & 15);
  float4 wv[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) wv[d] = *reinterpret_cast<const float4*>(W0 + (size_t) (4 * d) * N);
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int l = 0; l < LAYERS; ++l) {
    if (l >= layers) break;
    const float* Wn = W0 + (size_t) ((l + 1) % LAYERS) * N * N;
    v4f acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
    const float* hx = &h[cur][i_][b >> 2][0];
    float4 a4 = *reinterpret_cast<const float4*>(hx), nxt4 = a4;
#pragma unroll
    for (int d = 0; d < 64; ++d) {
      const float av = (d & 3) == 0 ? a4.x : (d & 3) == 1 ? a4.y : (d & 3) == 2 ? a4.z : a4.w;
      if ((d & 3) == 0 && d + 4 < 64) nxt4 = *reinterpret_cast<const float4*>(hx + d + 4);
      acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wv[d].w, acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      wv[d] = *reinterpret_cast<const float4*>(Wn + (size_t) (4 * d) * N);
      __builtin_amdgcn_sched_barrier(0);
      if ((d & 3) == 3) a4 = nxt4;
    }
    // sum over the four lane rows (r mod 4): 16 registers -> 8 -> 4
    float r[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) r[4 * q + i] = acc[q][i];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r[2 * p]), "+v"(r[2 * p + 1]));
      r[2 * p] += r[2 * p + 1];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r[4 * p]), "+v"(r[4 * p + 2]));
      r[4 * p] += r[4 * p + 2];
    }
    // register p = q: lanes 0-31 hold row i = 0 (lane row 0) / 2 (lane row 1) ... decoded by the probe
    if (decode) {
#pragma unroll
      for (int p = 0; p < 4; ++p) Y[(size_t) (wave * 64 + lane) * 4 + p] = r[4 * p];
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) h[cur ^ 1][p][lane & 3][16 * wave + (lane >> 2)] = fmaxf(r[4 * p], 0.f) * 0.1f;  // (timing: any bijection)
    }
    cur ^= 1;
    __syncthreads();
  }
  if (!decode) {
    float* Yb = Y + (size_t) blockIdx.x * ROWS * N;
    for (int t = threadIdx.x; t < ROWS * N; t += 256) Yb[t] = h[cur][t / N][(t % N) & 3][(t % N) >> 2];
  }
}

void run_quarter(const float* W, const float* X, float* Y, int nwg, int reps, const std::vector<float>& hW, const std::vector<float>& hX) {
  hipLaunchKernelGGL(layers_quarter_kernel, dim3(1), dim3(256), 0, 0, W, X, Y, 1, 1);
  std::vector<float> hT(1024);
  hipMemcpy(hT.data(), Y, 4096, hipMemcpyDeviceToHost);
  std::vector<double> ref(4 * N);
  for (int r = 0; r < 4; ++r)
    for (int n = 0; n < N; ++n) {
      double s2 = 0;
      for (int k = 0; k < N; ++k) s2 += (double) hX[r * N + k] * hW[(size_t) k * N + n];
      ref[r * N + n] = s2;
    }
  int found = 0;
  for (int L = 0; L < 256; ++L)
    for (int p = 0; p < 4; ++p) {
      const float v = hT[L * 4 + p];
      int hit = -1;
      for (int q = 0; q < 4 * N; ++q)
        if (fabs(ref[q] - v) < 2e-5 * (1 + fabs(v))) { hit = q; break; }
      if (hit >= 0) ++found;
      if (L < 64 && (L % 4 == 0 || L < 8)) printf("    lane %2d reg %d -> row %d output %3d\n", L, p, hit < 0 ? -1 : hit / N, hit < 0 ? -1 : hit % N);
    }
  printf("  quarter kernel: tree outputs matched %d of 1024\n", found);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(layers_quarter_kernel, dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS, 0);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(layers_quarter_kernel, dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("  4 rows/WG, weights [r][out], no split-K (a wave owns 64 outputs), a layer in flight: %7.2f  %6.2f\n", ms / reps * 1e3f,
         ms / reps * 1e3f / LAYERS);
}

template <int RG, bool REDUCE>
float time_it(const float* W, const float* X, float* Y, int nwg, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((layers_kernel<RG, REDUCE>), dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((layers_kernel<RG, REDUCE>), dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 64;
  const int rows_max = 8 * 128;
  std::vector<float> hW((size_t) LAYERS * N * N), hX((size_t) rows_max * N);
  srand(1);
  for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-4f;
  for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
  float *W, *X, *Y;
  hipMalloc(&W, hW.size() * 4), hipMalloc(&X, hX.size() * 4), hipMalloc(&Y, hX.size() * 4);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  // ---- layout check: one layer, 8 rows, one workgroup
  hipLaunchKernelGGL((layers_kernel<2, true>), dim3(1), dim3(256), 0, 0, W, X, Y, 1);
  std::vector<float> hY(8 * N);
  hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 8; ++r)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += (double) hX[r * N + k] * hW[(size_t) n * N + k];
      worst = fmax(worst, fabs(s - hY[r * N + n]));
    }
  printf("layout check (8 rows x 256 x 256, one layer): max abs err %.3g %s\n", worst, worst < 1e-4 ? "OK" : "MISMATCH");
  const int reps = 50;
  printf("%d workgroups, 8 layers of 256 x 256 per launch (us per launch / us per layer):\n", nwg);
  float t;
  t = time_it<1, true>(W, X, Y, nwg, reps);  printf("  4 rows/WG, block sums by shuffles : %7.2f  %6.2f\n", t, t / LAYERS);
  t = time_it<1, false>(W, X, Y, nwg, reps); printf("  4 rows/WG, no block sums          : %7.2f  %6.2f\n", t, t / LAYERS);
  t = time_it<2, true>(W, X, Y, nwg, reps);  printf("  8 rows/WG, block sums by shuffles : %7.2f  %6.2f\n", t, t / LAYERS);
  t = time_it<2, false>(W, X, Y, nwg, reps); printf("  8 rows/WG, no block sums          : %7.2f  %6.2f\n", t, t / LAYERS);
  if (nwg > 128) return 1;
  run_rowmajor<1, 4>(W, X, Y, nwg, reps, hW, hX);
  run_rowmajor<1, 8>(W, X, Y, nwg, reps, hW, hX);
  run_rowmajor<2, 8>(W, X, Y, nwg, reps, hW, hX);
  run_rowmajor<1, 8, 1>(W, X, Y, nwg, reps, hW, hX);
  run_rowmajor<2, 8, 1>(W, X, Y, nwg, reps, hW, hX);
  run_quarter(W, X, Y, nwg, reps, hW, hX);
  if (argc > 2) return 0;
  // ---- the pipelined kernel: where do the outputs land after the tree?  one layer, 4 rows, one workgroup
  hipLaunchKernelGGL(layers_pipelined_kernel, dim3(1), dim3(256), 0, 0, W, X, Y, 1, 1);
  std::vector<float> hT(1024);
  hipMemcpy(hT.data(), Y, 4096, hipMemcpyDeviceToHost);
  std::vector<double> ref(4 * N);
  for (int r = 0; r < 4; ++r)
    for (int n = 0; n < N; ++n) {
      double s2 = 0;
      for (int k = 0; k < N; ++k) s2 += (double) hX[r * N + k] * hW[(size_t) n * N + k];
      ref[r * N + n] = s2;
    }
  int found = 0;
  for (int w = 0; w < 1; ++w)
    for (int L = 0; L < 64; ++L)
      for (int p = 0; p < 4; ++p) {
        const float v = hT[(w * 64 + L) * 4 + p];
        int hit = -1;
        for (int q = 0; q < 4 * N; ++q)
          if (fabs(ref[q] - v) < 2e-5 * (1 + fabs(v))) { hit = q; break; }
        if (hit >= 0) ++found;
        if (L < 20 || L % 16 == 0) printf("    lane %2d reg %d -> row %d feature %3d\n", L, p, hit < 0 ? -1 : hit / N, hit < 0 ? -1 : hit % N);
      }
  printf("  tree outputs matched: %d of 256 (wave 0)\n", found);
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(layers_pipelined_kernel, dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS, 0);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(layers_pipelined_kernel, dim3(nwg), dim3(256), 0, 0, W, X, Y, LAYERS, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  4 rows/WG, pipelined (a layer in flight), transposing tree : %7.2f  %6.2f\n", ms / reps * 1e3f, ms / reps * 1e3f / LAYERS);
  }
  return 0;
}
