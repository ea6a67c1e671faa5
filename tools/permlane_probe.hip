// Probe for wave_sum9_transposed (skgs_common.h): checks the gfx950 permlane-swap semantics the reduction relies on.
// Build: hipcc --offload-arch=gfx950 -O2 -I sk_gs_amd/csrc tools/permlane_probe.hip -o tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "skgs_common.h"
__global__ void probe(const float* in, float* out) {
  const int lane = threadIdx.x;
  float v[9];
  for (int q = 0; q < 9; ++q) v[q] = in[q * 64 + lane];
  skgs::wave_sum9_transposed(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], 0xff00ff00ff00ff00ull);
  out[lane]      = v[1];
  out[64 + lane] = v[8];
}
int main() {
  float h[9 * 64], *d, *o, r[128];
  double want[9] = {0};
  for (int q = 0; q < 9; ++q)
    for (int l = 0; l < 64; ++l) h[q * 64 + l] = (float) ((q + 1) * 1000 + l * (q + 1)), want[q] += h[q * 64 + l];
  hipMalloc(&d, sizeof h), hipMalloc(&o, sizeof r);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 8; ++m) {
    int q = ((m & 1) << 2) | (m & 2) | ((m >> 2) & 1);
    printf("lane %2d: got %.1f want v%d = %.1f\n", 8 * m, r[8 * m], q, want[q]);
    bad += std::fabs(r[8 * m] - want[q]) > 0.5;
  }
  printf("lane 63: got %.1f want v8 = %.1f\n", r[64 + 63], want[8]);
  bad += std::fabs(r[64 + 63] - want[8]) > 0.5;
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  if (bad) for (int l = 0; l < 64; ++l) printf("%d:%.0f ", l, r[l]);
  return bad;
}
