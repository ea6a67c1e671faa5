// Probe for wave_sum9_banked (skgs_common.h): checks the gfx950 permlane-swap / bank-masked DPP semantics it relies on.
// Build: hipcc --offload-arch=gfx950 -O2 -I sk_gs_amd/csrc tools/permlane_probe.hip -o tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "skgs_common.h"
__global__ void probe(const float* in, float* out) {
  const int lane = threadIdx.x;
  float w[9];
  for (int q = 0; q < 9; ++q) w[q] = in[q * 64 + lane];
  skgs::wave_sum9_banked(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
  out[128 + lane] = w[0];
  out[192 + lane] = skgs::banked_holder(lane) ? (float) skgs::banked_holder_value(lane) : -1.f;
}
int main() {
  float h[9 * 64], *d, *o, r[256];
  double want[9] = {0};
  for (int q = 0; q < 9; ++q)
    for (int l = 0; l < 64; ++l) h[q * 64 + l] = (float) ((q + 1) * 1000 + l * (q + 1) + (l * l * (q + 3)) % 17), want[q] += h[q * 64 + l];
  hipMalloc(&d, sizeof h), hipMalloc(&o, sizeof r);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  int holders = 0;
  for (int l = 0; l < 64; ++l) {
    if (r[192 + l] < 0.f) continue;
    const int q = (int) r[192 + l];
    ++holders;
    printf("banked lane %2d: got %.1f want v%d = %.1f\n", l, r[128 + l], q, want[q]);
    bad += std::fabs(r[128 + l] - want[q]) > 0.5;
  }
  bad += holders != 9;
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  if (bad) for (int l = 0; l < 64; ++l) printf("%d:%.0f ", l, r[128 + l]);
  return bad;
}
