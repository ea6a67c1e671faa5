// render_blend.inl -- forward / backward blend kernels. Included TWICE by render.hip:
//   namespace blend_fast   : FMA contraction on, hardware exp (v_exp_f32)            -> the product path
//   namespace blend_strict : contraction off, the oracle's literal operation order and the reproducible
//                            double-precision exp of skgs_exp_strict()                  -> bit-comparable with the
//                            oracle built with exp_mode = 1 (parity tests; skgs_set_strict_math(1))
// SKGS_STRICT (0/1) and SKGS_BLEND_NS are defined by the includer.
namespace SKGS_BLEND_NS {

__device__ __forceinline__ float blend_exp(float x) {
#if SKGS_STRICT
  return skgs_exp_strict(x);
#else
  return __expf(x);
#endif
}
#if !SKGS_STRICT
// product build: the staged conic is pre-scaled by log2(e) as well, so the exponent goes straight into v_exp_f32 (2^x): one
// multiply less per visit in both kernels.  -DSKGS_BLEND_LOG2E_PRESCALE=0 builds the round-2 "e" form (conic scaled by
// -1/2 only, exp as v_mul + v_exp_f32) for A/B flip censuses (tools/flip_census_ab.sh).
#ifndef SKGS_BLEND_LOG2E_PRESCALE
#define SKGS_BLEND_LOG2E_PRESCALE 1
#endif
#if SKGS_BLEND_LOG2E_PRESCALE
constexpr float BLEND_LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float blend_exp2(float x_log2) { return __builtin_amdgcn_exp2f(x_log2); }
#else
constexpr float BLEND_LOG2E = 1.0f;
__device__ __forceinline__ float blend_exp2(float x) { return __expf(x); }
#endif
#endif

// ====================================================================================================== forward
// CENSUS (parity tests only, skgs_render_census): the same walk also leaves a fingerprint of WHICH list entries each pixel
// blended -- their number and the sum of census_mix(list position) -- so that a test can find exactly the pixels whose
// branch decisions differ from the oracle's.  The arithmetic is untouched (the census image is compared bit for bit with
// the product kernel's).
__device__ __forceinline__ uint32_t census_mix(uint32_t k) { return (k * 2654435761u) ^ (k >> 5); }
template <int PPL, int E, bool CENSUS = false>
__global__ void __launch_bounds__(64) render_forward_kernel(int W, int H, int gx, int T, TileRanges ranges,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ extra, const float* __restrict__ bg, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_opacity, float* __restrict__ out_extra,
    uint32_t* __restrict__ census = nullptr /* [H*W][2]: blended entries, sum of census_mix(position) */) {
  constexpr int SUBS = 4 / PPL;
  static_assert(XCD_GROUP == TILE_GROUP * 4, "one xcd_remap group = one group of tiles at one pixel per lane");
  int tile, sub;
  if (!blend_work_item<SUBS>(xcd_remap(blockIdx.x, T * SUBS), T, ranges.group_order, tile, sub)) return;
  const int lane = threadIdx.x;
  const Pix<PPL> pix = pixel_setup<PPL>(tile, sub, lane, gx, W, H);

  __shared__ float4 s_a[WAVE];  // x, y, conic a, conic b
  __shared__ float4 s_b[WAVE];  // conic c, opacity, r, g
  __shared__ float4 s_c[WAVE];  // b (.x): a 16-byte stride like s_a / s_b, so ONE address register serves the three reads
  __shared__ float s_e[E > 0 ? WAVE * E : 1];

  const int64_t start = ranges.begin[tile];
  const int64_t end   = min<int64_t>((int64_t) ranges.end[tile], capacity);
  const WaveRect rect = wave_rect<PPL>(tile, sub, gx);

  float Tr[PPL], C[PPL][3], Ex[PPL][E > 0 ? E : 1];
  uint32_t last[PPL];
  uint32_t cen_n[PPL], cen_h[PPL];
  bool done[PPL];
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    cen_n[i] = 0, cen_h[i] = 0;
    Tr[i] = 1.0f, last[i] = 0, done[i] = !pix.inside[i];
    C[i][0] = C[i][1] = C[i][2] = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) Ex[i][e] = 0.f;
  }

  // fast build, one pixel per lane: the finished pixels as a lane mask (the loop below keeps its decisions in SGPRs)
  constexpr bool LANE_MASKS = PPL == 1 && !SKGS_STRICT;
  unsigned long long done_m = __builtin_amdgcn_ballot_w64(done[0]);
  for (int64_t base = start; base < end; base += WAVE) {
    if (LANE_MASKS) {
      if (done_m == ~0ull) break;
    } else {
      bool all_done = true;
#pragma unroll
      for (int i = 0; i < PPL; ++i) all_done = all_done && done[i];
      if (__all(all_done)) break;
    }
    const int n = (int) min<int64_t>(WAVE, end - base);
    __syncthreads();  // single-wave workgroup: orders the LDS reads of the previous batch before these writes
    bool relevant = false;
    if (lane < n) {
      const uint32_t id = point_list[base + lane];
      const float4 a = recs[3 * id], b = recs[3 * id + 1], c = recs[3 * id + 2];
#if SKGS_STRICT
      s_a[lane] = a, s_b[lane] = b, s_c[lane].x = c.x;
#else
      // the staged record carries the conic pre-scaled: power log2(e) = dx (p dx + q dy) + (r dy) dy with p = -A/2 log2(e),
      // q = -B log2(e), r = -C/2 log2(e)
      s_a[lane] = make_float4(a.x, a.y, (-0.5f * BLEND_LOG2E) * a.z, -BLEND_LOG2E * a.w);
      s_b[lane] = make_float4((-0.5f * BLEND_LOG2E) * b.x, b.y, b.z, b.w), s_c[lane].x = c.x;
#endif
#pragma unroll
      for (int e = 0; e < E; ++e) s_e[lane * (E > 0 ? E : 1) + e] = extra[(size_t) id * E + e];
      relevant = splat_reaches_rect(a.x, a.y, a.z, a.w, b.x, c.w, rect);
    }
    __syncthreads();
    const uint32_t contrib0 = (uint32_t) (base - start);
    [[maybe_unused]] uint32_t lastj = 0xffffffffu;  // LANE_MASKS: slot of the lane's last contributing splat in THIS batch
    // only the splats whose 1/255 iso-contour can reach this wave's pixel rectangle are visited (wave-uniform list)
    // (the visited bit is cleared by ONE s_bitset0_b64 -- `todo &= todo - 1` is s_add_u32 / s_addc_u32 / s_and_b64: a wave issues
    // one instruction of any kind per turn of its SIMD, and the visit is a chain)
    for (unsigned long long todo = __ballot(relevant); todo;) {
      const int j    = __builtin_ctzll(todo);
      asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(j));
      const float4 a = s_a[j];
      const float4 b = s_b[j];
#if SKGS_STRICT
      // literal control flow and operation order of the reference (gaussian_render.cu:66-100)
      bool hit[PPL];
      float al[PPL], Tp[PPL];
      bool any = false;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        hit[i] = false;
        al[i] = 0.f, Tp[i] = 0.f;
        if (!done[i]) {
          const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          if (power <= 0.0f) {
            const float alpha = fminf(0.99f, b.y * blend_exp(power));
            if (alpha >= ALPHA_MIN) {
              const float test_T = Tr[i] * (1.f - alpha);
              if (test_T < T_MIN) {
                done[i] = true;
              } else {
                hit[i]  = true;
                al[i]   = alpha;
                Tp[i]   = Tr[i];
                Tr[i]   = test_T;
                last[i] = contrib0 + j + 1;
                if constexpr (CENSUS) cen_n[i] += 1, cen_h[i] += census_mix(contrib0 + j + 1);
              }
            }
          }
        }
        any = any || hit[i];
      }
      if (__ballot(any) != 0) {
        const float cb = s_c[j].x;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          // reference order: features * alpha * T, left to right (gaussian_render.cu:93-95)
          if (hit[i]) {
            C[i][0] += b.z * al[i] * Tp[i];
            C[i][1] += b.w * al[i] * Tp[i];
            C[i][2] += cb * al[i] * Tp[i];
#pragma unroll
            for (int e = 0; e < E; ++e) Ex[i][e] += s_e[j * (E > 0 ? E : 1) + e] * al[i] * Tp[i];
          }
        }
      }
#else
      // Same decisions, branch-free: after the wave-level cull nearly every visit has a contributing lane, so the
      // nested exec-mask regions only cost SALU work and serialise the LDS reads.  Lanes that do not contribute
      // carry weight 0 (adds an exact +0).
      const float cb = s_c[j].x;
      if constexpr (LANE_MASKS) {
        // one pixel per lane: the decisions live as lane masks in SGPRs -- `stop` and `hit` are the two halves of `valid`
        // under ONE compare (from the bool form below the compiler emits the compare and its complement), and the index of
        // the last contributing splat is moved under the hit mask instead of through a VGPR copy and a select: 24 VALU
        // instructions per visit instead of 26, 54.9 -> 51.8 us at config #1.  Round 4: a wave issues ONE instruction of any kind
        // per turn of its SIMD and the visit is a dependent chain, so its twenty scalar instructions count at low residency (the
        // launch's last third): the batch slot j itself is what moves (contrib0 + j + 1 is formed once per batch, behind the walk),
        // exec is saved and narrowed by one s_and_saveexec_b64, the visited bit cleared by one s_bitset0_b64: 46 -> 42
        // instructions per visit, 47.9 -> 46.0 us with the bit clear alone
        const float dx = a.x - pix.x[0], dy = a.y - pix.y[0];
        const float power  = dx * (a.z * dx + a.w * dy) + (b.x * dy) * dy;
        const float alpha  = fminf(0.99f, b.y * blend_exp2(power));
        const float test_T = Tr[0] * (1.f - alpha);
        const unsigned long long m_valid = ~done_m & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                           __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);
        const unsigned long long m_lt  = __builtin_amdgcn_ballot_w64(test_T < T_MIN);
        const unsigned long long m_hit = m_valid & ~m_lt, m_stop = m_valid & m_lt;
        const float aT = alpha * Tr[0];
        float wgt;
        asm("v_cndmask_b32_e64 %0, 0, %3, %5\n\t"
            "v_cndmask_b32_e64 %1, %1, %4, %5\n\t"
            "s_and_saveexec_b64 s[2:3], %5\n\t"
            "v_mov_b32_e32 %2, %6\n\t"
            "s_mov_b64 exec, s[2:3]"
            : "=&v"(wgt), "+v"(Tr[0]), "+v"(lastj)
            : "v"(aT), "v"(test_T), "s"(m_hit), "s"(j)
            : "s2", "s3", "scc");
        done_m |= m_stop;
        if constexpr (CENSUS) {
          const bool h = (m_hit >> lane) & 1ull;
          cen_n[0] += h ? 1u : 0u, cen_h[0] += h ? census_mix(contrib0 + j + 1) : 0u;
        }
        C[0][0] += b.z * wgt;
        C[0][1] += b.w * wgt;
        C[0][2] += cb * wgt;
#pragma unroll
        for (int e = 0; e < E; ++e) Ex[0][e] += s_e[j * (E > 0 ? E : 1) + e] * wgt;
        continue;
      }
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
        const float power  = dx * (a.z * dx + a.w * dy) + (b.x * dy) * dy;  // pre-scaled conic, see the staging
        const float alpha  = fminf(0.99f, b.y * blend_exp2(power));
        const float test_T = Tr[i] * (1.f - alpha);
        const bool valid   = !done[i] && power <= 0.0f && alpha >= ALPHA_MIN;
        const bool stop    = valid && test_T < T_MIN;
        const bool hit     = valid && !stop;
        const float wgt    = hit ? alpha * Tr[i] : 0.f;
        done[i]            = done[i] || stop;
        Tr[i]              = hit ? test_T : Tr[i];
        last[i]            = hit ? contrib0 + j + 1 : last[i];
        if constexpr (CENSUS) cen_n[i] += hit ? 1u : 0u, cen_h[i] += hit ? census_mix(contrib0 + j + 1) : 0u;
        C[i][0] += b.z * wgt;
        C[i][1] += b.w * wgt;
        C[i][2] += cb * wgt;
#pragma unroll
        for (int e = 0; e < E; ++e) Ex[i][e] += s_e[j * (E > 0 ? E : 1) + e] * wgt;
      }
#endif
    }
#if !SKGS_STRICT
    if constexpr (LANE_MASKS) last[0] = lastj != 0xffffffffu ? contrib0 + lastj + 1 : last[0];
#endif
  }
  const size_t HW = (size_t) H * W;
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    if (pix.inside[i]) {
      out_opacity[pix.id[i]] = 1.f - Tr[i];
      n_contrib[pix.id[i]]   = last[i];
      if constexpr (CENSUS) census[2 * (size_t) pix.id[i]] = cen_n[i], census[2 * (size_t) pix.id[i] + 1] = cen_h[i];
      // optional background: C + T * bg (upstream diff_gaussian_rasterization epilogue; NULL = in-tree variant)
      out_color[pix.id[i]]          = bg ? C[i][0] + Tr[i] * bg[0] : C[i][0];
      out_color[HW + pix.id[i]]     = bg ? C[i][1] + Tr[i] * bg[1] : C[i][1];
      out_color[2 * HW + pix.id[i]] = bg ? C[i][2] + Tr[i] * bg[2] : C[i][2];
#pragma unroll
      for (int e = 0; e < E; ++e) out_extra[e * HW + pix.id[i]] = Ex[i][e];
    }
  }
}

// ===================================================================================================== backward
// gradacc row layout (16 floats = 64 B per Gaussian):
//   0 mean2D.x  1 mean2D.y  2 conic.x  3 conic.y  4 conic.w  5 opacity  6..8 colour  9..12 extras  13..15 unused
// (fast build: slots 0..4 hold the moments sum w {dx, dy, dx^2, dx dy, dy^2} instead, see the kernel body)
template <int PPL, int E>
__global__ void __launch_bounds__(64) render_backward_kernel(int W, int H, int gx, int T, TileRanges ranges,
    int64_t capacity, const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
    const float* __restrict__ extra, const float* __restrict__ bg, const float* __restrict__ out_opacity,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_extra, const float* __restrict__ dL_dout_opacity /* may be NULL */,
    float* __restrict__ gradacc) {
  constexpr int SUBS = 4 / PPL;
  constexpr int NV   = 9 + E;
  int tile, sub;
  if (!blend_work_item<SUBS>(xcd_remap(blockIdx.x, T * SUBS), T, ranges.group_order, tile, sub)) return;
  const int lane = threadIdx.x;
  const Pix<PPL> pix = pixel_setup<PPL>(tile, sub, lane, gx, W, H);

  __shared__ float4 s_a[WAVE];
  __shared__ float4 s_b[WAVE];
  __shared__ float2 s_cid[2 * WAVE];  // [2 j]: (b, Gaussian id bits): 16-byte stride, one address register for the record
  __shared__ float s_e[E > 0 ? WAVE * E : 1];

  const int64_t start = ranges.begin[tile];
  const int64_t end   = min<int64_t>((int64_t) ranges.end[tile], capacity);
  const size_t HW     = (size_t) H * W;

  const WaveRect rect = wave_rect<PPL>(tile, sub, gx);
  float T_final[PPL], Tr[PPL], dL_dT[PPL], dpix[PPL][3], dex[PPL][E > 0 ? E : 1];
  // accum = colour blended behind the current splat.  The reference updates it lazily at the next contributor
  // (accum = last_alpha * last_color + (1 - last_alpha) * accum, gaussian_render.cu:281-287); here the same expression
  // is evaluated right after the contributor itself -- identical operands and rounding, 4 + E fewer state registers.
  float accum[PPL][3], accum_e[PPL][E > 0 ? E : 1];
  uint32_t lastk[PPL];
  uint32_t maxk = 0;
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    const bool in = pix.inside[i];
    T_final[i]    = in ? 1.0f - out_opacity[pix.id[i]] : 0.f;
    Tr[i]         = T_final[i];
    dL_dT[i]      = (in && dL_dout_opacity) ? -dL_dout_opacity[pix.id[i]] : 0.f;
    lastk[i]      = in ? n_contrib[pix.id[i]] : 0u;
    maxk          = max(maxk, lastk[i]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dpix[i][c]  = in ? dL_dpixels[c * HW + pix.id[i]] : 0.f;
      accum[i][c] = 0.f;
    }
    // out_color = C + T_final * bg: the background adds bg . dL_dpixel to dL/dT_final
    if (bg) dL_dT[i] += bg[0] * dpix[i][0] + bg[1] * dpix[i][1] + bg[2] * dpix[i][2];
#if !SKGS_STRICT
    dL_dT[i] *= -T_final[i];  // the walk only needs K = -T_final * dL/dT_final
#endif
#pragma unroll
    for (int e = 0; e < E; ++e) {
      dex[i][e]     = in ? dL_dout_extra[e * HW + pix.id[i]] : 0.f;
      accum_e[i][e] = 0.f;
    }
  }
  // wave-wide maximum of the last contributor: nothing behind it can matter to this wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) maxk = max(maxk, (uint32_t) __shfl_xor((int) maxk, d));
  maxk = (uint32_t) __builtin_amdgcn_readfirstlane((int) maxk);  // wave-uniform: the walk's bounds and k live in SGPRs
  if (maxk == 0) return;
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  const bool holder  = banked_holder(lane);
  const int holder_q = banked_holder_value(lane);

  // walk the list back to front: entry at list position k (0-based) has "contributor" index k
  for (int64_t hi = start + (int64_t) min<int64_t>(maxk, end - start); hi > start; hi -= WAVE) {
    const int n = (int) min<int64_t>(WAVE, hi - start);
    __syncthreads();
    bool relevant = false;
    if (lane < n) {
      const uint32_t id = point_list[hi - 1 - lane];
      const float4 a = recs[3 * id], b = recs[3 * id + 1], c = recs[3 * id + 2];
#if SKGS_STRICT
      s_a[lane] = a, s_b[lane] = b, s_cid[2 * lane] = make_float2(c.x, __uint_as_float(id));
#else
      s_a[lane] = make_float4(a.x, a.y, (-0.5f * BLEND_LOG2E) * a.z, -BLEND_LOG2E * a.w);
      s_b[lane] = make_float4((-0.5f * BLEND_LOG2E) * b.x, b.y, b.z, b.w);
      s_cid[2 * lane] = make_float2(c.x, __uint_as_float(id));
#endif
#pragma unroll
      for (int e = 0; e < E; ++e) s_e[lane * (E > 0 ? E : 1) + e] = extra[(size_t) id * E + e];
      relevant = splat_reaches_rect(a.x, a.y, a.z, a.w, b.x, c.w, rect);
    }
    __syncthreads();
    // (the forward's walk clears the bit with s_bitset0_b64; here, at eight waves per SIMD throughout and VALU-bound, that form
    // measured 1 us SLOWER)
    for (unsigned long long todo = __ballot(relevant); todo; todo &= todo - 1) {
      const int j      = __builtin_ctzll(todo);
      const uint32_t k = (uint32_t) (hi - 1 - j - start);
      const float4 a = s_a[j];
      const float4 b = s_b[j];
      float g[NV];
      bool any = false;
      float col[3] = {b.z, b.w, 0.f};
      bool col_loaded = false;
#if SKGS_STRICT
      // ---- the reference's expressions, term by term (gaussian_render.cu:252-318)
#pragma unroll
      for (int q = 0; q < NV; ++q) g[q] = 0.f;
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        if (k < lastk[i]) {
          const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
          const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
          if (power <= 0.0f) {
            const float G     = blend_exp(power);
            const float alpha = fminf(0.99f, b.y * G);
            if (alpha >= ALPHA_MIN) {
              any = true;
              if (!col_loaded) col[2] = s_cid[2 * j].x, col_loaded = true;
              const float Tn = Tr[i] / (1.f - alpha);
              const float tf_over = -T_final[i] / (1.f - alpha);
              Tr[i]          = Tn;
              const float dchannel_dcolor = alpha * Tn;
              float dL_dalpha = 0.0f;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                dL_dalpha += (col[c] - accum[i][c]) * dpix[i][c];
                g[6 + c] += dchannel_dcolor * dpix[i][c];
                accum[i][c] = alpha * col[c] + (1.f - alpha) * accum[i][c];
              }
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const float ce = s_e[j * (E > 0 ? E : 1) + e];
                dL_dalpha += (ce - accum_e[i][e]) * dex[i][e];
                g[9 + e] += dchannel_dcolor * dex[i][e];
                accum_e[i][e] = alpha * ce + (1.f - alpha) * accum_e[i][e];
              }
              dL_dalpha *= Tn;
              dL_dalpha += tf_over * dL_dT[i];
              const float dL_dG    = b.y * dL_dalpha;
              const float gdx      = G * dx;
              const float gdy      = G * dy;
              const float dG_ddelx = -gdx * a.z - gdy * a.w;
              const float dG_ddely = -gdy * b.x - gdx * a.w;
              g[0] += dL_dG * dG_ddelx * ddelx_dx;
              g[1] += dL_dG * dG_ddely * ddely_dy;
              g[2] += -0.5f * gdx * dx * dL_dG;
              g[3] += -0.5f * gdx * dy * dL_dG;
              g[4] += -0.5f * gdy * dy * dL_dG;
              g[5] += G * dL_dalpha;
            }
          }
        }
      }
#else
      // ---- product build.  Inside the divergent region only the per-pixel state and two scalars are produced:
      //   gA  = G * dL/dalpha                      (0 for a pixel the splat does not touch)
      //   dch = alpha * T_next = dC/dcolour        (0 likewise)
      // with  sum_c (colour_c - behind_c) dpix_c = D - S,  D = colour . dpix,  S = behind . dpix  (S obeys the recurrence
      // of the behind-colour: one state register instead of 3 + E)  and
      //   (D - S) T_next - T_final / (1 - alpha) dL/dT = ((D - S) T_prev + K) / (1 - alpha),  K = -T_final dL/dT.
      // The nine partials are plain products of gA / dch formed by ALL lanes afterwards (no zero-initialisation of
      // nine registers per visit): the five geometric ones are the moments of gA (w = o gA = G dL/dG without the
      // wave-uniform opacity o),
      //   sum gA {dx, dy, dx^2, dx dy, dy^2}
      // to which preprocess_backward applies the opacity and the conic coefficients once per Gaussian.
      float gA[PPL], dch[PPL], dxs[PPL], dys[PPL];
      unsigned long long any_mask = 0ull;
      const float2 cid = s_cid[2 * j];
      col[2] = cid.x;
      uint32_t gid = __float_as_uint(cid.y);  // (read here, with the record, not after the reduction where its latency is exposed)
      asm volatile("" : "+v"(gid));  // all three LDS reads of the record are issued together, none inside the divergent region
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        const float dx = a.x - pix.x[i], dy = a.y - pix.y[i];
        dxs[i] = dx, dys[i] = dy, gA[i] = 0.f, dch[i] = 0.f;
        float D = col[0] * dpix[i][0] + col[1] * dpix[i][1] + col[2] * dpix[i][2];  // colour . dL/dpixel (all lanes)
#pragma unroll
        for (int e = 0; e < E; ++e) D += s_e[j * (E > 0 ? E : 1) + e] * dex[i][e];
        asm volatile("" : "+v"(D));  // keep D (and the LDS reads behind it) in front of the branch
        // the three tests as one lane mask (the region below is entered by exactly the lanes the reference blends)
        const float power = dx * (a.z * dx + a.w * dy) + (b.x * dy) * dy;  // pre-scaled conic, see the staging
        const float G     = blend_exp2(power);
        const float alpha = fminf(0.99f, b.y * G);
        const bool valid  = (k < lastk[i]) & (power <= 0.0f) & (alpha >= ALPHA_MIN);
        // wave-level "does any lane blend this splat": the AND of the three compare masks, kept in SGPRs (a ballot of
        // the combined bool is lowered through a VGPR select + compare)
        any_mask |= __builtin_amdgcn_ballot_w64(k < lastk[i]) & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                    __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);
        if (valid) {
          const float rinv  = __builtin_amdgcn_rcpf(1.f - alpha);  // 1 ulp; the IEEE divide is ~10 instructions
          const float Tprev = Tr[i];
          const float Tn    = Tprev * rinv;
          Tr[i]             = Tn;
          const float t     = D - accum[i][0];
          const float dL_dalpha = (t * Tprev + dL_dT[i]) * rinv;
          accum[i][0] = fmaf(alpha, t, accum[i][0]);  // = alpha D + (1 - alpha) S, updated in place (one instruction, no copy)
          gA[i]  = G * dL_dalpha;
          dch[i] = alpha * Tn;
        }
      }
#endif
#if SKGS_STRICT
      const unsigned long long any_mask = __builtin_amdgcn_ballot_w64(any);
#endif
      if (any_mask != 0) {
#if !SKGS_STRICT
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          const float w  = gA[i];  // (x opacity = G dL/dG: applied once per Gaussian by preprocess_backward)
          const float m1 = w * dxs[i], m2 = w * dys[i];
          const float v[9] = {m1, m2, m1 * dxs[i], m1 * dys[i], m2 * dys[i], gA[i], dch[i] * dpix[i][0], dch[i] * dpix[i][1],
              dch[i] * dpix[i][2]};
#pragma unroll
          for (int q = 0; q < 9; ++q) g[q] = i == 0 ? v[q] : g[q] + v[q];
#pragma unroll
          for (int e = 0; e < E; ++e) g[9 + e] = i == 0 ? dch[i] * dex[i][e] : g[9 + e] + dch[i] * dex[i][e];
        }
#endif
        // the nine sums land in nine lanes of ONE register (banked_holder): ONE atomic instruction adds them into the
        // Gaussian's 64-B gradient row (a single memory-side request)
        wave_sum9_banked(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]);
#pragma unroll
        for (int q = 9; q < NV; ++q) g[q] = wave_sum_to_lane63(g[q]);
#if SKGS_STRICT
        const uint32_t gid = __float_as_uint(s_cid[2 * j].y);
#endif
        float* row = gradacc + (size_t) gid * GRAD_ROW;
#if SKGS_STRICT
        if (holder) atomicAdd(row + holder_q, g[0]);
#else
        // 32-bit byte offset from the (wave-uniform) table base: one v_lshl_add_u32 instead of a 64-bit shift and add per
        // visit (rows are 64 B: good for 2^26 Gaussians, checked by the host)
        if (holder) {
          const uint32_t off = (gid << 6) + (uint32_t) holder_q * 4u;
          atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(gradacc) + off), g[0]);
        }
#endif
        if (lane == 63) {
#pragma unroll
          for (int q = 9; q < NV; ++q) atomicAdd(row + q, g[q]);
        }
      }
    }
  }
}

}  // namespace SKGS_BLEND_NS
