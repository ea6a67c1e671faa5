// weight_stream_patterns.hip -- how fast can ONE workgroup (8 waves) pull a 256 x 256 fp32 matrix (256 KB) out of L2 into
// registers, by access pattern of a wave instruction?  (csrc/sp_mlp.hip: the forward's weight operand.)
//   A: 16 rows x 64 B per dwordx4 instruction (MFMA B-operand direct: lane (j, q) -> row j, bytes 64 s + 16 q)
//   B:  8 rows x 128 B (lane L -> row L / 8, bytes 128 s + 16 (L % 8))
//   C:  1 row x 1 KB  (lane L -> bytes 16 L of one row)
//   D: as A but dwordx2 along the row of a [k][f] layout: 4 rows x 128 B  (the backward's pattern)
// build: hipcc -O3 --offload-arch=gfx950 weight_stream_patterns.hip -o weight_stream_patterns ; run: ./weight_stream_patterns [nwg]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int N = 256, LAYERS = 8;

template <int PATTERN>
__global__ void __launch_bounds__(512) stream_kernel(const float* __restrict__ W, float* __restrict__ out, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int rep = 0; rep < reps; ++rep)
    for (int l = 0; l < LAYERS; ++l) {
      const float* Wl = W + (size_t) l * N * N;
      float4 b[32];
      if (PATTERN == 0) {  // wave owns rows 32 w .. 32 w + 31: tile c rows 32 w + 2 j + c
        const int j = lane & 15, q = lane >> 4;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int c = 0; c < 2; ++c) b[2 * s + c] = *reinterpret_cast<const float4*>(Wl + (size_t) (32 * wave + 2 * j + c) * N + 16 * s + 4 * q);
      } else if (PATTERN == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {  // 8 rows x 128 B per instruction: rows 32 w + 8 (i % 4) + L / 8, bytes 128 (i / 4)
          b[i] = *reinterpret_cast<const float4*>(Wl + (size_t) (32 * wave + 8 * (i & 3) + (lane >> 3)) * N + 32 * (i >> 2) + 4 * (lane & 7));
        }
      } else if (PATTERN == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) b[i] = *reinterpret_cast<const float4*>(Wl + (size_t) (32 * wave + i) * N + 4 * lane);
      } else {  // the backward: float2 at [k = 16 s + 4 q + t][32 w + 2 j]: 4 rows x 128 B, 64 instructions of 8 B per lane
        const int j = lane & 15, q = lane >> 4;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const float2 lo = *reinterpret_cast<const float2*>(Wl + (size_t) (8 * i + q) * N + 32 * wave + 2 * j);
          const float2 hi = *reinterpret_cast<const float2*>(Wl + (size_t) (8 * i + 4 + q) * N + 32 * wave + 2 * j);
          b[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc.x += b[i].x, acc.y += b[i].y, acc.z += b[i].z, acc.w += b[i].w;
      __syncthreads();
    }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[threadIdx.x] = acc.x;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 32;
  float *W, *out;
  hipMalloc(&W, sizeof(float) * LAYERS * N * N);
  hipMalloc(&out, 4096);
  std::vector<float> h(LAYERS * N * N, 1.0f);
  hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* names[4] = {"A 16 rows x 64 B (MFMA operand direct)", "B  8 rows x 128 B", "C  1 row x 1 KB", "D  4 rows x 128 B, dwordx2 (backward)"};
  for (int p = 0; p < 4; ++p) {
    auto launch = [&](int reps) {
      if (p == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(nwg), dim3(512), 0, 0, W, out, reps);
      if (p == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(512), 0, 0, W, out, reps);
      if (p == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(512), 0, 0, W, out, reps);
      if (p == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(nwg), dim3(512), 0, 0, W, out, reps);
    };
    launch(2);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    launch(reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-44s %4d workgroups: %7.1f us per 8-layer pass (2 MB per workgroup: %6.1f GB/s per CU)\n", names[p], nwg, us,
        LAYERS * N * N * 4 / us / 1e3);
  }
  return 0;
}
