// How long does the chip take to START a one-round grid?  A kernel whose every wave only waits `spin` clocks (s_memtime) is launched
// with the grid / block / LDS shapes of the per-Gaussian launches; time per launch (200 back-to-back launches between two events)
// minus the spin = launch + dispatch ramp + drain.   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_ramp tools/micro/dispatch_ramp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin_kernel(long long spin, float* out) {
  extern __shared__ float s[];
  const long long t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) s[0] = 1.f;
  while (__builtin_readcyclecounter() - t0 < spin) {}
  if (threadIdx.x == 0 && blockIdx.x == 0 && s[0] < 0.f) out[0] = 1.f;
}
int main() {
  float* out;
  hipMalloc(&out, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  struct Cfg { int grid, block, lds; const char* what; };
  const Cfg cfgs[] = {{391, 256, 50 * 1024, "preprocess_backward's shape (391 x 256 threads, 50 KB LDS)"},
                      {391, 256, 0, "the same without LDS"},
                      {782, 128, 24 * 1024, "preprocess_forward's shape (782 x 128 threads, 24 KB LDS)"},
                      {782, 128, 0, "the same without LDS"},
                      {1563, 64, 0, "one wave per workgroup (1563 x 64)"},
                      {98, 1024, 0, "98 x 1024 threads"},
                      {256, 1024, 20 * 1024, "scatter's shape (256 x 1024, 20 KB)"},
                      {2500, 256, 3 * 1024, "the blend kernels' shape (2500 x 256, 3 KB)"},
                      {1875, 256, 24 * 1024, "image_loss' shape (1875 x 256, 24 KB)"}};
  for (long long spin_us : {0LL, 5LL, 20LL}) {
    const long long spin = spin_us * 100;  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9
    printf("every wave waits %lld us\n", spin_us);
    for (const Cfg& c : cfgs) {
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin_kernel, dim3(c.grid), dim3(c.block), c.lds, 0, spin, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(spin_kernel, dim3(c.grid), dim3(c.block), c.lds, 0, spin, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("  %-62s %7.2f us per launch (%+6.2f over the wait)\n", c.what, ms * 5.f, ms * 5.f - (float) spin_us);
    }
  }
  return 0;
}
