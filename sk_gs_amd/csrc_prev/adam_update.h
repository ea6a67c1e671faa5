// adam_update.h -- the Adam update of one chunk of one tensor, shared by the optimizer's own launch (adam.hip) and the
// side job of the deform network's backward launch (mlp_fused.hip: the Gaussian rows' update runs on the CUs that launch
// leaves idle).  Math: see adam.hip.
#pragma once
#include <cstddef>
#include <cstdint>

#include "skgs_common.h"

namespace skgs {

struct AdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;        // elements
  int64_t chunk0;   // first chunk index of this tensor in the flattened chunk space
  float lr;
  int32_t sched;    // 0: the rate is `lr`; k > 0: slot k - 1 of the optimizer state's scheduled rates (skgs_adam_set_lr_schedules)
};
static_assert(sizeof(AdamTensor) == 56, "layout shared with the host binding");
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_CHUNK   = ADAM_THREADS * 4 * 4;  // elements per 256-thread group and iteration (4 float4 per lane)

// The optimizer's device state (skgs_adam_state_bytes() = 256, zero-initialised = "no step taken"): the step count and
// q_k = 1 - beta_k^count as doubles, advanced by recurrence (q' = (1 - beta) + beta q) -- a double pow() per launch was
// 2-3 us in front of every update; and, in its own 128-byte line, the ticket of the launches that advance the count
// themselves (last workgroup out).
// Learning-rate schedules on the device (the reference calls update_learning_rate before EVERY train step: train.py:140-141,
// networks/gaussian_splatting.py:56-84,455-470, networks/sk_gs.py:611-632): up to 4 schedules (ADAM_MAX_SCHEDULES), each `get_expon_lr_func`'s
// parameters; the launch that advances the step counter evaluates them for the step that follows and keeps the closed step's
// rates for the pieces that still belong to it (after_advance).  A step replayed inside a hipGraph -- four steps per replay in
// bench.py -- therefore follows the reference's rate step for step, with no host in the loop.
constexpr int ADAM_MAX_SCHEDULES = 4;
struct LrSchedule {  // include/skgs.h::skgs_lr_schedule
  double lr_init, lr_final, lr_delay_mult;
  double log_lr_init, log_lr_final;  // numpy's np.log of the two rates, formed by the HOST (the reference's own arithmetic)
  int32_t lr_delay_steps, max_steps, step_offset, reserved;
};
static_assert(sizeof(LrSchedule) == 56, "layout shared with the host binding");
struct AdamState {
  float count;       // steps taken so far
  float pad0;
  double q1, q2;     // 1 - beta1^count, 1 - beta2^count
  unsigned n_sched;  // byte 24
  unsigned pad_a;
  float lr_now[ADAM_MAX_SCHEDULES];   // byte 32: the scheduled rates of training step count + 1 (the step in progress)
  float lr_prev[ADAM_MAX_SCHEDULES];  // byte 48: ... of step `count` (pieces that run after the advance but belong to it)
  float lr_next[ADAM_MAX_SCHEDULES];  // byte 64: ... of step count + 2, staged early by the advancing launch (off its tail)
  const LrSchedule* sched;            // byte 80
  float pad1[10];
  unsigned ticket;   // word 32
  unsigned pad2[31];
};
static_assert(sizeof(AdamState) == 256, "layout shared with the host binding");
static_assert(offsetof(AdamState, lr_now) == 32 && offsetof(AdamState, sched) == 80 && offsetof(AdamState, ticket) == 128, "layout");

struct AdamCoef {
  float bc1, inv_sqrt_bc2, beta1, beta2, omb1, omb2, eps;
  float lr_sched[ADAM_MAX_SCHEDULES];
};
// get_expon_lr_func's helper (gaussian_splatting.py:69-82) in the double arithmetic numpy gives it, for training step `step`
// (1-based: the reference passes global_step + 1 / self._step); the float the optimizer applies is its rounding
__device__ __forceinline__ float lr_schedule_eval(const LrSchedule& s, int step) {
  step -= s.step_offset;
  if (step < 0 || (s.lr_init == 0.0 && s.lr_final == 0.0)) return 0.f;
  double delay_rate = 1.0;
  if (s.lr_delay_steps > 0) {
    const double u = fmin(fmax((double) step / (double) s.lr_delay_steps, 0.0), 1.0);
    delay_rate = s.lr_delay_mult + (1.0 - s.lr_delay_mult) * sin(0.5 * 3.14159265358979323846 * u);
  }
  const double t = fmin(fmax((double) step / (double) s.max_steps, 0.0), 1.0);
  return (float) (delay_rate * exp(s.log_lr_init * (1.0 - t) + s.log_lr_final * t));
}
// The advancing launch, in two parts.  EARLY (any one wave, before its own work; `count` = the counter as the launch found it): the
// rates of the step after next into the staging slots -- nobody reads those during the launch, and the double exp / sin are off the
// launch's tail.  LATE (the thread(s) that move the counter, after every reader of lr_now is done): now -> prev, staged -> now.
__device__ __forceinline__ void lr_schedules_stage(AdamState* st, int lane, float count) {
  const int n = (int) st->n_sched;
  if (lane < n && lane < ADAM_MAX_SCHEDULES) st->lr_next[lane] = lr_schedule_eval(st->sched[lane], (int) count + 2);
}
__device__ __forceinline__ void lr_schedules_commit(AdamState* st, int lane) {
  if (lane < ADAM_MAX_SCHEDULES) {
    st->lr_prev[lane] = st->lr_now[lane];
    st->lr_now[lane]  = st->lr_next[lane];
  }
}
// bias corrections of step count + 1.  Hyper-parameters arrive as doubles and (1 - beta) is formed in double, as torch
// does: 1.0f - 0.999f is off by 1.3e-5
// after_advance: the piece runs AFTER the launch that advanced the counter but belongs to that step (a side job of the next
// skeleton-forward launch): its bias corrections are the state's own 1 - beta^count.
__device__ __forceinline__ AdamCoef adam_coefficients(double beta1d, double beta2d, float eps, const AdamState* st,
    bool after_advance = false) {
  const double bc1 = after_advance ? st->q1 : (1.0 - beta1d) + beta1d * st->q1;
  const double bc2 = after_advance ? st->q2 : (1.0 - beta2d) + beta2d * st->q2;
  AdamCoef k{(float) bc1, (float) (1.0 / sqrt(bc2)), (float) beta1d, (float) beta2d, (float) (1.0 - beta1d),
      (float) (1.0 - beta2d), eps, {}};
#pragma unroll
  for (int i = 0; i < ADAM_MAX_SCHEDULES; ++i) k.lr_sched[i] = after_advance ? st->lr_prev[i] : st->lr_now[i];
  return k;
}
// the rate a tensor's update uses in this launch: its own, or its schedule slot's
__device__ __forceinline__ float adam_lr(const AdamTensor& T, const AdamCoef& k) {
  float lr = T.lr;
#pragma unroll
  for (int i = 0; i < ADAM_MAX_SCHEDULES; ++i) lr = T.sched == i + 1 ? k.lr_sched[i] : lr;
  return lr;
}
// (lanes of ONE wave: lane 0 moves the counter, the schedules advance one lane each)
__device__ __forceinline__ void adam_advance(AdamState* st, double beta1d, double beta2d, int lane = 0) {
  const float count = st->count;
  lr_schedules_stage(st, lane, count);
  if (lane == 0) {
    st->count = count + 1.0f;
    st->q1 = (1.0 - beta1d) + beta1d * st->q1;
    st->q2 = (1.0 - beta2d) + beta2d * st->q2;
  }
  lr_schedules_commit(st, lane);  // (same lane wrote the staging slot it reads)
}

// which tensor owns a chunk: lane i keeps the first chunk of tensor i (loaded once by the caller into `first0`, INT64_MAX
// beyond the table); the owner is the number of tensors whose first chunk is <= chunk, minus one.  (A linear walk over the
// descriptors was a chain of dependent global loads per chunk -- ~30 of them for the tensors at the end of the table.)
__device__ __forceinline__ int adam_owner(const AdamTensor* __restrict__ tensors, int n_tensors, int64_t first0, int lane,
    int64_t chunk) {
  int ti = __popcll(__ballot(first0 <= chunk)) - 1;
  for (int base = 64; base < n_tensors; base += 64)  // (more than 64 tensors: rare)
    ti += __popcll(__ballot(base + lane < n_tensors && tensors[base + lane].chunk0 <= chunk));
  return __builtin_amdgcn_readfirstlane(ti);
}

// The descriptors of tensors 0..63, one per lane (13 dwords), loaded up front beside the state: the owner's descriptor is
// then read out of the lanes (v_readlane with the wave-uniform owner) instead of by a second dependent global load.
struct AdamTensorLanes {
  uint32_t w[14];
};
__device__ __forceinline__ AdamTensorLanes adam_load_descriptors(const AdamTensor* __restrict__ tensors, int n_tensors, int lane) {
  AdamTensorLanes d;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(tensors + (lane < n_tensors ? lane : 0));
#pragma unroll
  for (int i = 0; i < 14; ++i) d.w[i] = p[i];
  return d;
}
__device__ __forceinline__ AdamTensor adam_descriptor_of(const AdamTensorLanes& d, int ti) {
  uint32_t w[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) w[i] = (uint32_t) __builtin_amdgcn_readlane((int) d.w[i], ti);
  AdamTensor T;
  __builtin_memcpy(&T, w, sizeof(T));
  return T;
}

// Every rounding is spelled out (no a * b + c left for the compiler to contract one way here and another way there): the
// update runs in several kernels -- the optimizer's own launch, the side jobs of the network launches, the sparse logit-table
// update -- and they must agree to the bit (a step taken in pieces, a sparse step against a dense one).  The form is the one
// the optimizer's launch had been compiled to: m' = fma(beta1, m, (1 - beta1) g), v' = fma(beta2, v, ((1 - beta2) g) g),
// p' = p - (lr / bc1 m') / fma(sqrt(v'), 1 / sqrt(bc2), eps).
__device__ __forceinline__ void adam_update_element(float& p, float& m, float& v, float g, float step_size, const AdamCoef& k) {
  m = __builtin_fmaf(k.beta1, m, __fmul_rn(k.omb1, g));
  v = __builtin_fmaf(k.beta2, v, __fmul_rn(__fmul_rn(k.omb2, g), g));
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(step_size, m), __builtin_fmaf(sqrtf(v), k.inv_sqrt_bc2, k.eps)));
}

// one chunk (ADAM_CHUNK elements from `base`) of tensor T, by 256 threads; t256 = this thread's index among them
__device__ __forceinline__ void adam_update_chunk(const AdamTensor& T, int64_t base, int t256, const AdamCoef& k) {
  const float step_size = adam_lr(T, k) / k.bc1;
  const bool aligned = ((reinterpret_cast<uintptr_t>(T.param) | reinterpret_cast<uintptr_t>(T.grad) |
                         reinterpret_cast<uintptr_t>(T.exp_avg) | reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15) == 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + ((int64_t) r * ADAM_THREADS + t256) * 4;
    if (aligned && i + 3 < T.n) {
      const float4 g = stream_load4<NT_ADAM_LOAD>(T.grad + i);
      float4 m = stream_load4<NT_ADAM_LOAD>(T.exp_avg + i);
      float4 v = stream_load4<NT_ADAM_LOAD>(T.exp_avg_sq + i);
      float4 p = stream_load4<NT_ADAM_LOAD>(T.param + i);
      adam_update_element(p.x, m.x, v.x, g.x, step_size, k);
      adam_update_element(p.y, m.y, v.y, g.y, step_size, k);
      adam_update_element(p.z, m.z, v.z, g.z, step_size, k);
      adam_update_element(p.w, m.w, v.w, g.w, step_size, k);
      stream_store4<NT_ADAM>(T.exp_avg + i, m);
      stream_store4<NT_ADAM>(T.exp_avg_sq + i, v);
      stream_store4<NT_ADAM>(T.param + i, p);
    } else {
      for (int64_t e = i; e < T.n && e < i + 4; ++e) {
        float p = T.param[e], m = T.exp_avg[e], v = T.exp_avg_sq[e];
        adam_update_element(p, m, v, T.grad[e], step_size, k);
        T.exp_avg[e] = m, T.exp_avg_sq[e] = v, T.param[e] = p;
      }
    }
  }
}

// how many chunks a 256-thread half of a side-job workgroup takes per iteration (mlp_fused.hip, sp_mlp.hip)
#ifndef SKGS_SIDE_CHUNKS
#define SKGS_SIDE_CHUNKS 1
#endif
// two chunks at once (any two tensors): all 32 loads of a thread are issued before the first update
__device__ __forceinline__ void adam_update_chunk2(const AdamTensor& A, int64_t baseA, const AdamTensor& Bt, int64_t baseB, int t256,
    const AdamCoef& k) {
  const bool al = ((reinterpret_cast<uintptr_t>(A.param) | reinterpret_cast<uintptr_t>(A.grad) | reinterpret_cast<uintptr_t>(A.exp_avg) |
                    reinterpret_cast<uintptr_t>(A.exp_avg_sq) | reinterpret_cast<uintptr_t>(Bt.param) | reinterpret_cast<uintptr_t>(Bt.grad) |
                    reinterpret_cast<uintptr_t>(Bt.exp_avg) | reinterpret_cast<uintptr_t>(Bt.exp_avg_sq)) & 15) == 0;
  const bool full = baseA + ADAM_CHUNK <= A.n && baseB + ADAM_CHUNK <= Bt.n;
  if (!(al && full)) {
    adam_update_chunk(A, baseA, t256, k);
    adam_update_chunk(Bt, baseB, t256, k);
    return;
  }
  float4 g[2][4], m[2][4], v[2][4], p[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const AdamTensor& T = c ? Bt : A;
    const int64_t base  = c ? baseB : baseA;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = base + ((int64_t) r * ADAM_THREADS + t256) * 4;
      g[c][r] = stream_load4<NT_ADAM_LOAD>(T.grad + i);
      m[c][r] = stream_load4<NT_ADAM_LOAD>(T.exp_avg + i);
      v[c][r] = stream_load4<NT_ADAM_LOAD>(T.exp_avg_sq + i);
      p[c][r] = stream_load4<NT_ADAM_LOAD>(T.param + i);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const AdamTensor& T = c ? Bt : A;
    const int64_t base  = c ? baseB : baseA;
    const float ss      = adam_lr(T, k) / k.bc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = base + ((int64_t) r * ADAM_THREADS + t256) * 4;
      adam_update_element(p[c][r].x, m[c][r].x, v[c][r].x, g[c][r].x, ss, k);
      adam_update_element(p[c][r].y, m[c][r].y, v[c][r].y, g[c][r].y, ss, k);
      adam_update_element(p[c][r].z, m[c][r].z, v[c][r].z, g[c][r].z, ss, k);
      adam_update_element(p[c][r].w, m[c][r].w, v[c][r].w, g[c][r].w, ss, k);
      stream_store4<NT_ADAM>(T.exp_avg + i, m[c][r]);
      stream_store4<NT_ADAM>(T.exp_avg_sq + i, v[c][r]);
      stream_store4<NT_ADAM>(T.param + i, p[c][r]);
    }
  }
}

}  // namespace skgs
