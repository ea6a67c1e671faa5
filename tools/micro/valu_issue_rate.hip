// What one SIMD of gfx950 issues per clock, measured: wave64 instructions of a few kinds in long independent chains, enough
// waves per SIMD to hide every latency.  Prints cycles per wave-instruction per SIMD -- the VALU ceiling that the roofline of
// the blend kernels is priced against (bench.py, `roofline.from_profile.valu`).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_rate tools/micro/valu_issue_rate.hip && ./valu_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#include <vector>

constexpr int ITERS = 4096, UNROLL = 16;

template <int KIND>
__global__ void __launch_bounds__(256) chain_kernel(float* out, float seed) {
  float a[UNROLL];
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p[UNROLL];
  for (int i = 0; i < UNROLL; ++i) a[i] = seed + i + threadIdx.x, p[i] = float2v{a[i], a[i] + 1.f};
  const float m = 1.0000001f, c = 1e-9f;
  unsigned long long mask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
  int sg = 0;
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(m) : "vcc");
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(float2v{m, m}), "v"(float2v{c, c}));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % UNROLL]));
      if (KIND == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 8) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mask));
      if (KIND == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % UNROLL]));
      if (KIND == 12) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % UNROLL]));
      if (KIND == 13) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 14) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 16) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 17) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i]));
      if (KIND == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
      if (KIND == 21) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[i]), "+s"(mask) : "v"(m));
      if (KIND == 22) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
      if (KIND == 23 && (i & 3) == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(m) : "vcc");
      if (KIND == 24 && (i & 3) == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(m) : "vcc");
      if (KIND == 18) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[(i + 1) % UNROLL]), "v"(m));
      // encodings: the same operation as VOP2 / VOPC (32-bit) and as VOP3 (64-bit)
      if (KIND == 30) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 31) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 32) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 33) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 34) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(a[i]), "v"(m));
      if (KIND == 35) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 36) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 37) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
      if (KIND == 38) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 39) asm volatile("v_max_f32_e32 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 40) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 41) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
    }
  }
  float s = 0;
  for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s + sg;
}

template <int KIND>
double run(const char* name, float* d_out, int cus) {
  const int waves_per_simd = 8, blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  chain_kernel<KIND><<<blocks, 256>>>(d_out, 1.f);
  hipEventRecord(e0);
  chain_kernel<KIND><<<blocks, 256>>>(d_out, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double insts_per_simd = (double) ITERS * UNROLL * waves_per_simd;
  const double cycles = ms * 1e-3 * khz * 1e3;
  printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %d MHz)  %7.1f G wave-instructions/s on %d CUs\n", name,
         ms, cycles / insts_per_simd, khz / 1000, insts_per_simd * 4 * cus / (ms * 1e-3) / 1e9, cus);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* d_out;
  hipMalloc(&d_out, 4);
  run<0>("v_fma_f32", d_out, cus);
  run<5>("v_mul_f32", d_out, cus);
  run<1>("v_pk_fma_f32", d_out, cus);
  run<2>("v_exp_f32", d_out, cus);
  run<3>("v_add_f32 dpp quad_perm", d_out, cus);
  run<8>("v_mov_b32 dpp row_shr", d_out, cus);
  run<4>("v_cndmask_b32", d_out, cus);
  run<7>("v_cmp_lt_f32", d_out, cus);
  run<9>("v_cndmask_b32 (sgpr mask)", d_out, cus);
  run<18>("v_cndmask_b32 (3 registers)", d_out, cus);
  run<19>("v_cndmask_b32_e64 vcc", d_out, cus);
  run<20>("v_cmp vcc + v_cndmask vcc (pair)", d_out, cus);
  run<22>("v_cmp vcc + s_nop + v_cndmask", d_out, cus);
  run<21>("v_cmp sgpr + v_cndmask sgpr", d_out, cus);
  run<23>("(v_cmp + 4 v_cndmask e32) / 4", d_out, cus);
  run<24>("(v_cmp + 4 v_cndmask e64) / 4", d_out, cus);
  run<6>("v_permlane32_swap_b32", d_out, cus);
  run<12>("v_permlane16_swap_b32", d_out, cus);
  run<13>("v_mov_b32 dpp row_bcast:31", d_out, cus);
  run<10>("v_add_f32", d_out, cus);
  run<16>("v_max_f32", d_out, cus);
  run<11>("v_mov_b32", d_out, cus);
  run<15>("v_add_u32", d_out, cus);
  run<14>("v_rcp_f32", d_out, cus);
  run<17>("v_readlane_b32", d_out, cus);
  run<30>("v_max_f32_e64", d_out, cus);
  run<41>("v_max_f32_e32 (operands swapped)", d_out, cus);
  run<39>("v_max_f32_e32 v, v, v (same reg)", d_out, cus);
  run<31>("v_min_f32_e32", d_out, cus);
  run<40>("v_med3_f32", d_out, cus);
  run<32>("v_add_f32_e64", d_out, cus);
  run<38>("v_mul_f32_e64", d_out, cus);
  run<35>("v_sub_f32_e32", d_out, cus);
  run<33>("v_fmac_f32_e32", d_out, cus);
  run<34>("v_cmp_lt_f32_e64 -> sgpr", d_out, cus);
  run<36>("v_and_b32_e32", d_out, cus);
  run<37>("v_lshlrev_b32_e32", d_out, cus);
  return 0;
}
