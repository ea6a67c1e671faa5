// adam.hip -- multi-tensor Adam step in ONE launch (scope row (f)-2).
//
// Reference: torch.optim.Adam(eps=1e-15, betas=(0.9, 0.999)) over the six Gaussian parameter groups with per-group
// learning rates (exps/default.yaml:122-125, networks/gaussian_splatting.py:443-453), stepped by
// my_ext/framework.py:264-306.  torch's fused path issues one multi_tensor_apply launch per group and state list
// (7 launches, ~300 us per step for config #1 on MI355X = 0.5 TB/s).  The update is a pure stream:
// 16 B read + 12 B written per element.  Here every tensor of every group is walked by one grid with float4 accesses;
// the step counter lives on the device so the launch can sit inside a captured hipGraph.
//
// Math (identical to torch, amsgrad = False, weight_decay = 0, maximize = False):
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <algorithm>
#include <cstdint>

#include "adam_update.h"
#include "skgs_common.h"

namespace skgs {
namespace {

// Optional first phase of ONE workgroup: the frequency-encoding backward (mlp.hip::freq_encode_backward_kernel,
// freqencoder.cu:36-60) that completes the gradient of the tensor whose (single) chunk is `chunk` -- the joint positions,
// whose gradient through the network input would otherwise need a launch of its own between backward and update.
struct FreqJob {
  int B, D, deg, ldo, accumulate;
  const float* g;
  const float* out;
  float* gx;
  int64_t chunk;  // the chunk whose workgroup runs the job first (-1: no job)
};
__device__ __forceinline__ void run_freq_job(const FreqJob& job) {
  for (int t = threadIdx.x; t < job.B * job.D; t += ADAM_THREADS) {
    const int b = t / job.D, d = t - b * job.D;
    const float* gr = job.g + (size_t) b * job.ldo;
    const float* o  = job.out + (size_t) b * job.ldo;
    float r = gr[d];
    for (int f = 0; f < job.deg; ++f) {
      const int s = job.D + 2 * f * job.D;
      r += scalbnf(1.0f, f) * (gr[s + d] * o[s + job.D + d] - gr[s + job.D + d] * o[s + d]);
    }
    job.gx[t] = job.accumulate ? job.gx[t] + r : r;
  }
  __syncthreads();  // (workgroup scope: the update below reads what this workgroup just wrote)
}

// Optional last act of the launch that closes a step: the NEXT view's record into the live view slot (sk_gs_amd/
// view_slot.py) -- slot[0..words) = table[order[cursor % n]], cursor += 1 -- so that a training loop that walks its views in
// a known order needs no copy between two replays of its graph (4.5 us per step as a device-to-device copy).
struct ViewAdvance {
  const uint32_t* table;  // [views][words]
  const int32_t* order;   // [n]
  int32_t* cursor;        // device counter
  uint32_t* slot;
  int n, words;
};
// advance = 1 (small grids only): the last workgroup out moves the counter and clears `zero_after` itself.  What it does there is
// stores only: the state's words were read with the coefficients at the start, and the view advance -- which depends on nothing
// this launch computes -- is the job of ONE EXTRA workgroup (the grid's last, no chunk of its own) that walks its chain of three
// dependent loads (cursor -> order -> record) beside the others' update instead of behind the ticket: 10.1 -> ~7 us.
__global__ void __launch_bounds__(ADAM_THREADS) adam_step_kernel(int n_tensors, const AdamTensor* __restrict__ tensors,
    int64_t chunk_begin, int64_t total_chunks, double beta1d, double beta2d, float eps, AdamState* __restrict__ state,
    int advance, float* __restrict__ zero_after, int64_t zero_n, FreqJob job, ViewAdvance va) {
  const int lane = threadIdx.x & 63;
  // one round trip: state, first chunks and descriptors are independent loads
  const AdamTensorLanes desc = adam_load_descriptors(tensors, n_tensors, lane);
  const int64_t first0 = lane < n_tensors ? tensors[lane].chunk0 : INT64_MAX;
  const AdamCoef k = adam_coefficients(beta1d, beta2d, eps, state);
  double pf_q1 = 0.0, pf_q2 = 0.0;
  float pf_count = 0.f;
  if (advance) pf_q1 = state->q1, pf_q2 = state->q2, pf_count = state->count;
  // the scheduled rates of the step after next, staged by the first wave of workgroup 0 while the update streams (nobody reads the
  // staging slots in this launch; its ticket below publishes them to the workgroup that commits)
  if (advance && blockIdx.x == 0 && threadIdx.x < 64) lr_schedules_stage(state, threadIdx.x, pf_count);
  if (advance && va.slot && blockIdx.x == gridDim.x - 1) {  // the extra workgroup: next view's record -> the live slot
    const int c = va.cursor[0];
    const int n = va.n > 0 ? va.n : max(va.cursor[1], 1);  // n_order = 0: the order's length is a device word too
    const uint32_t* rec = va.table + (size_t) va.order[c % n] * va.words;
    for (int i = threadIdx.x; i < va.words; i += ADAM_THREADS) va.slot[i] = rec[i];
    __syncthreads();  // every thread has read the cursor
    if (threadIdx.x == 0) va.cursor[0] = c + 1;
  }
  for (int64_t chunk = chunk_begin + blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    if (chunk == job.chunk) run_freq_job(job);
    const int ti = adam_owner(tensors, n_tensors, first0, lane, chunk);
    const AdamTensor T = ti < 64 ? adam_descriptor_of(desc, ti) : tensors[ti];
    adam_update_chunk(T, (chunk - T.chunk0) * ADAM_CHUNK, threadIdx.x, k);
  }
  if (advance) {
    __shared__ unsigned s_last;
    __syncthreads();  // every thread's reads of the state and of its gradients are done
    if (threadIdx.x == 0) {
      __threadfence();  // (workgroup 0's staged rates before its ticket)
      s_last = atomicAdd(&state->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
      for (int64_t i = threadIdx.x; i < zero_n; i += ADAM_THREADS) zero_after[i] = 0.f;
      if (threadIdx.x == 0) {  // adam_advance's expressions on the words read at the start (nobody has written them since)
        state->count  = pf_count + 1.0f;
        state->q1     = (1.0 - beta1d) + beta1d * pf_q1;
        state->q2     = (1.0 - beta2d) + beta2d * pf_q2;
        state->ticket = 0u;
      }
      __threadfence();
      lr_schedules_commit(state, threadIdx.x);  // now -> prev, staged -> now (lanes 0..3)
    }
  }
}

// (For large grids a last-workgroup-out ticket was measured instead of this launch: 4096 same-address atomics next to
// the counter every workgroup reads cost 70 us.)
// ... and, in the same launch, clears `zero_after` (gradient storage that must read zero when the next backward starts:
// the per-frame tables of which a step writes one row -- instead of a fill launch at the start of every step)
__global__ void __launch_bounds__(256) adam_bump_kernel(AdamState* state, double beta1d, double beta2d,
    float* __restrict__ zero_after, int64_t zero_n) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 64) adam_advance(state, beta1d, beta2d, threadIdx.x);
  if (i < zero_n) zero_after[i] = 0.f;
}


// installs the schedules in the optimizer's state and evaluates them for the state's CURRENT count: lr_now = the rates of training
// step count + 1, lr_prev = those of step count (also the way to re-derive them after a restored step count)
__global__ void __launch_bounds__(64) adam_set_schedules_kernel(AdamState* state, const LrSchedule* sched, int n) {
  const int lane = threadIdx.x;
  if (lane == 0) state->n_sched = (unsigned) n, state->sched = n > 0 ? sched : nullptr;
  if (lane < ADAM_MAX_SCHEDULES) {
    const int count = (int) state->count;
    state->lr_now[lane]  = lane < n ? lr_schedule_eval(sched[lane], count + 1) : 0.f;
    state->lr_prev[lane] = lane < n ? lr_schedule_eval(sched[lane], count) : 0.f;
    state->lr_next[lane] = lane < n ? lr_schedule_eval(sched[lane], count + 2) : 0.f;
  }
}

// ---- LBS_method 'W' (networks/sk_gs.py:469-471, exps/default.yaml:35): Adam on the dense [P, M] logit table, sparsely ----------
// A row of the table receives a gradient at its K nearest superpoints only (the gather of sk_gs.py:769): every other entry has
// g = 0, and as long as its moments are zero too its update is EXACTLY zero (m' = v' = 0, p' = p - lr/bc1 * 0 / (0 + eps)).
// So the dense update (1.4 GB of traffic per step at P = 100k, M = 512: 245 us, after a 200 MB dense gradient was written)
// only has to visit the 32-column tiles of a row that have EVER been touched: `tile_mask[row]`, one bit per tile, grows by the
// tiles of this step's neighbours and can be rebuilt from the moments at any time (skgs_adam_logit_mask_rebuild).  Inside a
// visited tile every element takes the full update (stale moments keep decaying, as in the dense launch): bit-identical
// parameters and moments.  The dense gradient is never formed: g = w_k (g_w_k - sum_j w_j g_w_j) at column indices[k]
// (the arithmetic of deform.hip::lbs_logits_dense_wide_kernel), zero elsewhere.
constexpr int LOGIT_TILE = 32;
__global__ void __launch_bounds__(256) adam_logit_rows_kernel(int P, int M, int K, const float* __restrict__ weights,
    const int64_t* __restrict__ indices, const float* __restrict__ g_weights, const AdamTensor* __restrict__ desc,
    uint32_t* __restrict__ tile_mask, double beta1d, double beta2d, float eps, const AdamState* __restrict__ state, int after_advance) {
  const int lane = threadIdx.x & 63, half = lane >> 5, e = lane & 31;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = gridDim.x * 4;
  const AdamCoef k = adam_coefficients(beta1d, beta2d, eps, state, after_advance != 0);
  const AdamTensor T    = desc[0];
  const float step_size = adam_lr(T, k) / k.bc1;
  for (int n = wave; n < P; n += n_waves) {
    // (n is wave-uniform: the K triples are scalar loads)
    float gl[16];
    int id[16];
    float dot = 0.f;
    for (int q = 0; q < K; ++q) dot = __builtin_fmaf(weights[(size_t) n * K + q], g_weights[(size_t) n * K + q], dot);
    uint32_t touched = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < K) {
        id[q] = (int) indices[(size_t) n * K + q];
        gl[q] = __fmul_rn(weights[(size_t) n * K + q], __fsub_rn(g_weights[(size_t) n * K + q], dot));
        touched |= 1u << (id[q] / LOGIT_TILE);
      } else {
        id[q] = -1, gl[q] = 0.f;
      }
    }
    const uint32_t before = tile_mask[n];
    uint32_t todo = __builtin_amdgcn_readfirstlane(before | touched);
    if (lane == 0 && todo != before) tile_mask[n] = todo;
    while (todo) {  // two tiles per pass: lanes 0-31 the lowest set bit, lanes 32-63 the next one
      const int ta = __builtin_ctz(todo);
      todo &= todo - 1;
      const int tb = todo ? __builtin_ctz(todo) : -1;
      if (todo) todo &= todo - 1;
      const int t = half ? tb : ta;
      const int j = t * LOGIT_TILE + e;
      if (t >= 0 && j < M) {
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (id[q] == j) g += gl[q];
        const size_t at = (size_t) n * M + j;
        float p = T.param[at], m = T.exp_avg[at], v = T.exp_avg_sq[at];
        adam_update_element(p, m, v, g, step_size, k);
        T.exp_avg[at] = m, T.exp_avg_sq[at] = v, T.param[at] = p;
      }
    }
  }
}
// The same update with R rows of a wave in flight at once (K <= KM): the triples and masks of R rows are requested together (scalar
// loads: the row numbers are made wave-uniform for the compiler), then the first tile pair of every row, and only a row with more than
// two live tiles takes further passes.  Same arithmetic per element.  Measured in the superpoint stage's step with LBS_method W
// (three alternating runs each): one row at a time 0.4668 ms, R = 2 0.4635, R = 4 0.4733 (91 registers: 5 waves per SIMD) -- the launch
// is close to what its 128-byte accesses stream at (~270 MB in 62 us), not a chain of round trips.
template <int KM, int R>
__global__ void __launch_bounds__(256) adam_logit_rows_batched_kernel(int P, int M, int K, const float* __restrict__ weights,
    const int64_t* __restrict__ indices, const float* __restrict__ g_weights, const AdamTensor* __restrict__ desc,
    uint32_t* __restrict__ tile_mask, double beta1d, double beta2d, float eps, const AdamState* __restrict__ state, int after_advance) {
  const int lane = threadIdx.x & 63, half = lane >> 5, e = lane & 31;
  // (readfirstlane: the row numbers are wave-uniform, which the compiler cannot see through threadIdx.x >> 6 -- the triples become
  // scalar loads into SGPRs instead of 64 copies in vector registers)
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), n_waves = gridDim.x * 4;
  const AdamCoef k = adam_coefficients(beta1d, beta2d, eps, state, after_advance != 0);
  const AdamTensor T    = desc[0];
  const float step_size = adam_lr(T, k) / k.bc1;
  for (int n0 = wave * R; n0 < P; n0 += n_waves * R) {
    // ---- every row's triples and mask (wave-uniform addresses; rows beyond P repeat the last one and do nothing)
    float w[R][KM], gw[R][KM];
    int id[R][KM];
    uint32_t before[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const size_t row = (size_t) min(n0 + r, P - 1) * K;
#pragma unroll
      for (int q = 0; q < KM; ++q) {
        const int qc = min(q, K - 1);
        w[r][q] = weights[row + qc], gw[r][q] = g_weights[row + qc], id[r][q] = (int) indices[row + qc];
      }
      before[r] = tile_mask[min(n0 + r, P - 1)];
    }
    float gl[R][KM];
    uint32_t todo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < KM; ++q)
        if (q < K) dot = __builtin_fmaf(w[r][q], gw[r][q], dot);
      uint32_t touched = 0u;
#pragma unroll
      for (int q = 0; q < KM; ++q) {
        if (q < K) {
          gl[r][q] = __fmul_rn(w[r][q], __fsub_rn(gw[r][q], dot));
          touched |= 1u << (id[r][q] / LOGIT_TILE);
        } else {
          id[r][q] = -1, gl[r][q] = 0.f;
        }
      }
      const bool valid = n0 + r < P;
      todo[r] = valid ? __builtin_amdgcn_readfirstlane(before[r] | touched) : 0u;
      if (lane == 0 && valid && todo[r] != before[r]) tile_mask[n0 + r] = todo[r];
    }
    // ---- the first tile pair of every row: all loads, then all updates
    float pp[R], mm[R], vv[R];
    size_t at[R];
    int jj[R];
    bool act[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t td = todo[r];
      const int ta = td ? __builtin_ctz(td) : -1;
      if (td) td &= td - 1;
      const int tb = td ? __builtin_ctz(td) : -1;
      if (td) td &= td - 1;
      todo[r] = td;
      const int t = half ? tb : ta;
      jj[r]  = t * LOGIT_TILE + e;
      act[r] = t >= 0 && jj[r] < M;
      at[r]  = act[r] ? (size_t) (n0 + r) * M + jj[r] : 0;
      pp[r] = T.param[at[r]], mm[r] = T.exp_avg[at[r]], vv[r] = T.exp_avg_sq[at[r]];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (act[r]) {
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (id[r][q] == jj[r]) g += gl[r][q];
        adam_update_element(pp[r], mm[r], vv[r], g, step_size, k);
        T.exp_avg[at[r]] = mm[r], T.exp_avg_sq[at[r]] = vv[r], T.param[at[r]] = pp[r];
      }
    }
    // ---- rows with more than two live tiles
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t td = todo[r];
      while (td) {
        const int ta = __builtin_ctz(td);
        td &= td - 1;
        const int tb = td ? __builtin_ctz(td) : -1;
        if (td) td &= td - 1;
        const int t = half ? tb : ta;
        const int j = t * LOGIT_TILE + e;
        if (t >= 0 && j < M) {
          float g = 0.f;
#pragma unroll
          for (int q = 0; q < KM; ++q)
            if (id[r][q] == j) g += gl[r][q];
          const size_t a = (size_t) (n0 + r) * M + j;
          float p = T.param[a], m = T.exp_avg[a], v = T.exp_avg_sq[a];
          adam_update_element(p, m, v, g, step_size, k);
          T.exp_avg[a] = m, T.exp_avg_sq[a] = v, T.param[a] = p;
        }
      }
    }
  }
}
// The same sparse visit for an optimizer that is NOT this package's: the dense gradient EXISTS (torch's autograd built it, or
// skgs_lbs_weights_backward wrote it) and is what the update reads -- the rule that makes skipping exact is the same: an element whose
// gradient and both moments are zero does not move.  tile_mask[row] grows by the tiles of `indices` [P,K] (the caller's knowledge of where
// this step's gradient can be non-zero; may be NULL) and, with `scan`, by every tile of the row that holds a non-zero gradient (one pass
// over the dense gradient: for a step in which something else than the known neighbours may have written to it).  Inside a live tile every
// element takes adam_update_element -- the dense launch's arithmetic: bit-identical parameters and moments.
__global__ void __launch_bounds__(256) adam_masked_rows_kernel(int P, int M, int K, const int64_t* __restrict__ indices, int scan,
    const AdamTensor* __restrict__ desc, uint32_t* __restrict__ tile_mask, double beta1d, double beta2d, float eps,
    const AdamState* __restrict__ state, int after_advance) {
  const int lane = threadIdx.x & 63, half = lane >> 5, e = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), n_waves = gridDim.x * 4;
  const AdamCoef k = adam_coefficients(beta1d, beta2d, eps, state, after_advance != 0);
  const AdamTensor T    = desc[0];
  const float step_size = adam_lr(T, k) / k.bc1;
  for (int n = wave; n < P; n += n_waves) {
    uint32_t touched = 0u;
    if (indices)
      for (int q = 0; q < K; ++q) {
        const long long j = indices[(size_t) n * K + q];
        if (j >= 0 && j < M) touched |= 1u << ((int) j / LOGIT_TILE);
      }
    if (scan)
      for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long b = __ballot(j < M && T.grad[(size_t) n * M + j] != 0.f);
        if (b & 0xffffffffull) touched |= 1u << (j0 / LOGIT_TILE);
        if (b >> 32) touched |= 1u << (j0 / LOGIT_TILE + 1);
      }
    const uint32_t before = tile_mask[n];
    uint32_t todo = __builtin_amdgcn_readfirstlane(before | touched);
    if (lane == 0 && todo != before) tile_mask[n] = todo;
    while (todo) {  // two tiles per pass: lanes 0-31 the lowest set bit, lanes 32-63 the next one
      const int ta = __builtin_ctz(todo);
      todo &= todo - 1;
      const int tb = todo ? __builtin_ctz(todo) : -1;
      if (todo) todo &= todo - 1;
      const int t = half ? tb : ta;
      const int j = t * LOGIT_TILE + e;
      if (t >= 0 && j < M) {
        const size_t at = (size_t) n * M + j;
        float p = T.param[at], m = T.exp_avg[at], v = T.exp_avg_sq[at];
        adam_update_element(p, m, v, T.grad[at], step_size, k);
        T.exp_avg[at] = m, T.exp_avg_sq[at] = v, T.param[at] = p;
      }
    }
  }
}

// a tile is live when any of its moments is non-zero (after a restore, a re-ordering or a change of the row count)
__global__ void __launch_bounds__(256) adam_logit_mask_kernel(int P, int M, const float* __restrict__ exp_avg,
    const float* __restrict__ exp_avg_sq, uint32_t* __restrict__ tile_mask) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = gridDim.x * 4;
  for (int n = wave; n < P; n += n_waves) {
    uint32_t mask = 0u;
    for (int j0 = 0; j0 < M; j0 += 64) {
      const int j = j0 + lane;
      const bool live = j < M && (exp_avg[(size_t) n * M + j] != 0.f || exp_avg_sq[(size_t) n * M + j] != 0.f);
      const unsigned long long b = __ballot(live);
      if (b & 0xffffffffull) mask |= 1u << (j0 / LOGIT_TILE);
      if (b >> 32) mask |= 1u << (j0 / LOGIT_TILE + 1);
    }
    if (lane == 0) tile_mask[n] = mask;
  }
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

size_t skgs_adam_tensor_bytes(void) { return sizeof(AdamTensor); }
int64_t skgs_adam_chunk_elems(void) { return ADAM_CHUNK; }
size_t skgs_adam_state_bytes(void) { return sizeof(AdamState); }

/* tensors: DEVICE array of n_tensors descriptors {param, grad, exp_avg, exp_avg_sq, n, chunk0, lr} (56 B each, every
 * pointer 16-B aligned; chunk0 = running sum of ceil(n / skgs_adam_chunk_elems())). step_state: the optimizer's device
 * state (skgs_adam_state_bytes(), zero-initialised; word 0 = the number of steps taken as a float); advanced by the call. */
int skgs_adam_step(int32_t n_tensors, const void* tensors, int64_t total_chunks, double beta1, double beta2, double eps,
    float* step_state, float* zero_after, int64_t zero_n, skgs_stream_t stream) {
  if (n_tensors == 0 || total_chunks == 0) return 0;
  return skgs_adam_step_range(n_tensors, tensors, 0, total_chunks, beta1, beta2, eps, step_state, 1, zero_after, zero_n,
      stream);
}

/* Learning-rate schedules evaluated on the device (the reference's per-iteration update_learning_rate: train.py:140-141,
 * gaussian_splatting.py:56-84,455-470, sk_gs.py:611-632).  schedules: DEVICE array of n <= 8 skgs_lr_schedule (it must stay alive and
 * unchanged while the optimizer steps; n <= 4); tensors whose descriptor carries slot k (the int32 after `lr`) take schedule k - 1's rate. */
int skgs_adam_set_lr_schedules(float* step_state, const skgs_lr_schedule* schedules, int32_t n, skgs_stream_t stream) {
  SKGS_REQUIRE(step_state && n >= 0 && n <= ADAM_MAX_SCHEDULES && (n == 0 || schedules), "adam_set_lr_schedules: 0 <= n <= 4 schedules");
  SKGS_REQUIRE((reinterpret_cast<uintptr_t>(step_state) & 7) == 0, "adam_set_lr_schedules: the state must be 8-byte aligned");
  static_assert(sizeof(skgs_lr_schedule) == sizeof(LrSchedule), "skgs_lr_schedule");
  hipLaunchKernelGGL(adam_set_schedules_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, reinterpret_cast<AdamState*>(step_state),
      reinterpret_cast<const LrSchedule*>(schedules), n);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {
int step_range_impl(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1, double beta2,
    double eps, float* step_state, int32_t advance, float* zero_after, int64_t zero_n, const FreqJob& job,
    const ViewAdvance& va, skgs_stream_t stream) {
  SKGS_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors) && step_state, "adam_step: NULL argument");
  SKGS_REQUIRE(chunk_begin >= 0 && chunk_end >= chunk_begin, "adam_step: bad chunk range");
  SKGS_REQUIRE((reinterpret_cast<uintptr_t>(step_state) & 7) == 0, "adam_step: the state must be 8-byte aligned");
  hipStream_t s    = (hipStream_t) stream;
  AdamState* state = reinterpret_cast<AdamState*>(step_state);
  const int64_t zn = zero_after ? std::max<int64_t>(zero_n, 0) : 0;
  const int64_t nc = n_tensors > 0 ? chunk_end - chunk_begin : 0;
  const bool self_advance = advance && nc > 0 && nc <= 256 && zn <= 65536;
  SKGS_REQUIRE(!va.slot || self_advance, "adam_step_tail: the view advance rides on a short closing piece (<= 256 chunks)");
  ProfScope prof(K_ADAM, s);
  if (nc > 0) {
    // (+ 1: the view advance's own workgroup; self_advance grids are one chunk per workgroup, so it finds no chunk)
    const int grid = (int) std::min<int64_t>(nc, 256 * 16) + ((self_advance && va.slot) ? 1 : 0);
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(ADAM_THREADS), 0, s, n_tensors,
        reinterpret_cast<const AdamTensor*>(tensors), chunk_begin, chunk_end, beta1, beta2, (float) eps, state,
        self_advance ? 1 : 0, zero_after, zn, job, va);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  if (advance && !self_advance) {
    hipLaunchKernelGGL(adam_bump_kernel, dim3((unsigned) std::max<int64_t>(1, (zn + 255) / 256)), dim3(256), 0, s, state,
        beta1, beta2, zero_after, zn);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
}  // namespace

/* One step taken in pieces: the chunks [chunk_begin, chunk_end) of the table (whole tensors: the chunk0 of a tensor and of
 * the one after it) are updated with the bias correction of step count + 1; the counter moves (and zero_after is
 * cleared) only where `advance` is set -- in the LAST piece, ordered after all the others.  Pieces of one step may run on
 * different streams, beside the backward kernels that do not touch their tensors.  chunk_begin == chunk_end with
 * advance = 1 only moves the counter.  (A short advancing piece does so itself, last workgroup out; a long one is followed
 * by a one-workgroup launch.) */
int skgs_adam_step_range(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1,
    double beta2, double eps, float* step_state, int32_t advance, float* zero_after, int64_t zero_n, skgs_stream_t stream) {
  FreqJob job{};
  job.chunk = -1;
  return step_range_impl(n_tensors, tensors, chunk_begin, chunk_end, beta1, beta2, eps, step_state, advance, zero_after, zero_n,
      job, ViewAdvance{}, stream);
}

/* The closing piece of a step (advance = 1) whose range holds a tensor with an unfinished gradient: the workgroup that
 * owns chunk `freq_chunk` (the single chunk of that small tensor: the joint positions) first runs
 * skgs_freq_encode_backward(freq_B, freq_D, freq_degree, freq_grad_out, freq_out, freq_ld_out, freq_grad_x,
 * freq_accumulate) itself -- gradient completed and consumed without a launch in between. */
int skgs_adam_step_tail(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1,
    double beta2, double eps, float* step_state, float* zero_after, int64_t zero_n, int64_t freq_chunk, int32_t freq_B,
    int32_t freq_D, int32_t freq_degree, const float* freq_grad_out, const float* freq_out, int32_t freq_ld_out,
    float* freq_grad_x, int32_t freq_accumulate, const skgs_view_advance* next_view, skgs_stream_t stream) {
  FreqJob job{};
  job.chunk = -1;
  ViewAdvance va{};
  if (next_view && next_view->slot) {
    SKGS_REQUIRE(next_view->table && next_view->order && next_view->cursor && next_view->n_order >= 0 && next_view->words >= 1,
        "adam_step_tail: bad view advance");
    va = ViewAdvance{reinterpret_cast<const uint32_t*>(next_view->table), next_view->order, next_view->cursor,
        reinterpret_cast<uint32_t*>(next_view->slot), next_view->n_order, next_view->words};
  }
  if (freq_grad_x) {
    SKGS_REQUIRE(freq_B >= 0 && freq_D >= 1 && freq_degree >= 0 && freq_grad_out && freq_out, "adam_step_tail: bad encoder job");
    SKGS_REQUIRE(freq_chunk >= chunk_begin && freq_chunk < chunk_end, "adam_step_tail: the job's chunk is outside the range");
    SKGS_REQUIRE((int64_t) freq_B * freq_D <= skgs_adam_chunk_elems(), "adam_step_tail: the job's tensor must be one chunk");
    job = FreqJob{freq_B, freq_D, freq_degree, freq_ld_out, freq_accumulate, freq_grad_out, freq_out, freq_grad_x, freq_chunk};
  }
  return step_range_impl(n_tensors, tensors, chunk_begin, chunk_end, beta1, beta2, eps, step_state, 1, zero_after, zero_n, job,
      va, stream);
}

/* Adam on the dense [P, M] LBS-logit table of LBS_method 'W' without the dense gradient (see adam_logit_rows_kernel): the update
 * of skgs_adam_step_range(advance = 0) for that ONE tensor -- `tensor` = its descriptor in the optimizer's DEVICE table
 * (param, exp_avg, exp_avg_sq, lr, n = P * M are read from it; its grad pointer is not) -- given the step's K neighbours per row
 * and the cotangent of their softmax weights; bit-identical parameters and moments.  tile_mask [P] uint32: persistent, zero at
 * the start of training (all moments zero), else skgs_adam_logit_mask_rebuild.  M <= 1024, K <= 16. */
int skgs_adam_logit_rows(int32_t P, int32_t M, int32_t K, const float* weights, const int64_t* indices, const float* g_weights,
    const void* tensor, uint32_t* tile_mask, double beta1, double beta2, double eps, const float* step_state, int32_t after_advance,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && M >= 1 && M <= 32 * LOGIT_TILE && K >= 1 && K <= 16 && K <= M, "adam_logit_rows: need M <= 1024, 1 <= K <= min(16, M)");
  if (P == 0) return 0;
  SKGS_REQUIRE(weights && indices && g_weights && tensor && tile_mask && step_state, "adam_logit_rows: NULL argument");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_ADAM, s);
  static const int rows_in_flight = [] {  // SKGS_LOGIT_ROWS=1: the one-row-at-a-time kernel (A/B measurements)
    const char* e_ = getenv("SKGS_LOGIT_ROWS");
    return e_ ? atoi(e_) : 2;
  }();
  if (K <= 8 && rows_in_flight == 2) {
    const int grid = (int) std::min<int64_t>(((int64_t) P + 7) / 8, 256 * 8);
    hipLaunchKernelGGL((adam_logit_rows_batched_kernel<8, 2>), dim3(grid), dim3(256), 0, s, P, M, K, weights, indices, g_weights,
        reinterpret_cast<const AdamTensor*>(tensor), tile_mask, beta1, beta2, (float) eps, reinterpret_cast<const AdamState*>(step_state),
        after_advance ? 1 : 0);
  } else {
    const int grid = (int) std::min<int64_t>(((int64_t) P + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(adam_logit_rows_kernel, dim3(grid), dim3(256), 0, s, P, M, K, weights, indices, g_weights,
        reinterpret_cast<const AdamTensor*>(tensor), tile_mask, beta1, beta2, (float) eps, reinterpret_cast<const AdamState*>(step_state),
        after_advance ? 1 : 0);
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
int skgs_adam_masked_rows(int32_t P, int32_t M, int32_t K, const int64_t* indices, int32_t scan_gradient, const void* tensor,
    uint32_t* tile_mask, double beta1, double beta2, double eps, const float* step_state, int32_t after_advance, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && M >= 1 && M <= 32 * LOGIT_TILE && K >= 0 && K <= 64, "adam_masked_rows: need M <= 1024");
  if (P == 0) return 0;
  SKGS_REQUIRE(tensor && tile_mask && step_state && (indices || K == 0), "adam_masked_rows: NULL argument");
  hipStream_t s = (hipStream_t) stream;
  ProfScope prof(K_ADAM, s);
  const int grid = (int) std::min<int64_t>(((int64_t) P + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(adam_masked_rows_kernel, dim3(grid), dim3(256), 0, s, P, M, K, indices, scan_gradient ? 1 : 0,
      reinterpret_cast<const AdamTensor*>(tensor), tile_mask, beta1, beta2, (float) eps, reinterpret_cast<const AdamState*>(step_state),
      after_advance ? 1 : 0);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
int skgs_adam_logit_mask_rebuild(int32_t P, int32_t M, const float* exp_avg, const float* exp_avg_sq, uint32_t* tile_mask,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && M >= 1 && M <= 32 * LOGIT_TILE, "adam_logit_mask_rebuild: need M <= 1024");
  if (P == 0) return 0;
  SKGS_REQUIRE(exp_avg && exp_avg_sq && tile_mask, "adam_logit_mask_rebuild: NULL argument");
  hipStream_t s = (hipStream_t) stream;
  const int grid = (int) std::min<int64_t>(((int64_t) P + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(adam_logit_mask_kernel, dim3(grid), dim3(256), 0, s, P, M, exp_avg, exp_avg_sq, tile_mask);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
