// densify.hip -- per-iteration densification statistics (scope row (f)-4, first part).
//
// Reference, run after every backward of the densification phase (networks/sk_gs.py:1990-1997,
// networks/gaussian_splatting.py:503-513):
//     mask = radii > 0
//     max_radii2D[mask]        = max(max_radii2D[mask], radii[mask])
//     xyz_gradient_accum[mask] += || viewspace_points.grad[mask, :2] ||_2
//     denom[mask]              += 1
// i.e. three masked index_put / index_select round trips plus a norm (~10 torch kernels).  One streaming kernel here:
// 20 B read + 12 B written per Gaussian.
#include <algorithm>

#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int32_t* __restrict__ radii,
    const float* __restrict__ grad_means2D /*[P,3]*/, float mult, float* __restrict__ xyz_gradient_accum,
    float* __restrict__ denom, float* __restrict__ max_radii2D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r <= 0) return;
  const float gx = grad_means2D[3 * i], gy = grad_means2D[3 * i + 1];
  max_radii2D[i]        = fmaxf(max_radii2D[i], (float) r);
  const float nrm       = sqrtf(gx * gx + gy * gy);
  xyz_gradient_accum[i] = xyz_gradient_accum[i] + (mult == 1.0f ? nrm : mult * nrm);
  denom[i]              = denom[i] + 1.0f;
}

// ---- densification surgery (scope row (f)-4, second part): every per-Gaussian tensor AND its two Adam moments rebuilt
// by ONE launch.  Reference: change_optimizer / prune_points / densification_postfix (networks/gaussian_splatting.py:
// 515-587) index or concatenate each of the 7 parameters, exp_avg and exp_avg_sq one by one (~100 torch launches, several
// host synchronisations per clone / split / prune).  Every such operation is a row gather:
//   dst_t[i, :] = src_t[rows[i], :]               for i < n_keep   (surviving Gaussians keep parameter and moments)
//   dst_t[i, :] = src_t[rows[i], :]  or  0        for i >= n_keep  (new Gaussians copy their parent's parameters; their
//                                                                   moments start from zero, :531-545)
struct RowTensor {
  const float* src;
  float* dst;
  int32_t row_floats;
  int32_t fresh_is_zero;
};
static_assert(sizeof(RowTensor) == 24, "descriptor layout shared with sk_gs_amd/optim.py");

__global__ void __launch_bounds__(256) gather_rows_kernel(const RowTensor* __restrict__ tensors, long long n_out, long long n_keep,
    const int64_t* __restrict__ rows) {
  const RowTensor t = tensors[blockIdx.y];
  const long long total = n_out * t.row_floats;
  const long long stride = (long long) gridDim.x * 256;
  for (long long e = (long long) blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const long long i = e / t.row_floats;
    const int c       = (int) (e - i * t.row_floats);
    t.dst[e] = (i >= n_keep && t.fresh_is_zero) ? 0.f : t.src[rows[i] * t.row_floats + c];
  }
}


// ---- the decisions of adaptive density control on the device (scope row (f)-4).  Reference: densify_and_clone /
// densify_and_split / prune (networks/gaussian_splatting.py:589-650) evaluate their masks with ~25 torch kernels and turn
// them into index lists with boolean indexing (a host synchronisation per mask).  Here: one launch writes a flag byte per
// Gaussian, one single-workgroup launch compacts the flagged rows into the row list of skgs_gather_rows, group after group,
// in index order -- the host reads back the group sizes once (it has to: the new tensors are sized by them).
constexpr int FLAG_KEEP = 1, FLAG_CLONE = 2, FLAG_SPLIT = 4;

__global__ void __launch_bounds__(256) densify_flags_kernel(int P, const float* __restrict__ accum, const float* __restrict__ denom,
    const float* __restrict__ log_scale, float max_grad, float scene_extent, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float g = accum[i] / denom[i];
  if (g != g) g = 0.f;                                    // grads[grads.isnan()] = 0 (:638)
  const float s = fmaxf(fmaxf(expf(log_scale[3 * i]), expf(log_scale[3 * i + 1])), expf(log_scale[3 * i + 2]));
  const bool big   = s > scene_extent;
  const bool clone = fabsf(g) >= max_grad && !big;        // torch.norm(grads, dim=-1) >= t  and  max scale <= extent (:624-626)
  const bool split = g >= max_grad && big;                // padded grads >= t  and  max scale > extent (:595-597)
  flags[i] = (uint8_t) ((split ? 0 : FLAG_KEEP) | (clone ? FLAG_CLONE : 0) | (split ? FLAG_SPLIT : 0));
}

__global__ void __launch_bounds__(256) prune_flags_kernel(int P, const float* __restrict__ opacity_logit,
    const float* __restrict__ max_radii2D /* NULL: no screen / world size tests */, const float* __restrict__ log_scale,
    float min_opacity, float max_screen_size, float world_size_limit, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  bool drop = 1.0f / (1.0f + expf(-opacity_logit[i])) < min_opacity;   // (:645)
  if (max_radii2D) {
    const float s = fmaxf(fmaxf(expf(log_scale[3 * i]), expf(log_scale[3 * i + 1])), expf(log_scale[3 * i + 2]));
    drop = drop || max_radii2D[i] > max_screen_size || s > world_size_limit;   // (:647-649)
  }
  flags[i] = drop ? 0 : FLAG_KEEP;
}

// rows = [i : flags[i] & bit_0] ++ [i : flags[i] & bit_1] ++ ... , group g repeated repeat_g times as whole blocks (the
// `.repeat(N)` of :599-606); counts[g] = size of group g (one block).  Three short launches: per-tile counts (a tile =
// 2048 consecutive Gaussians, 8 per thread), a one-workgroup scan of the tile counts, and the ordered scatter (block scan
// of the per-thread counts + the tile's offset).  (One workgroup sweeping all P flags twice: 0.5 ms at 300k.)
constexpr int COMPACT_GROUPS = 4, TILE_T = 256, TILE_E = 8, CTILE = TILE_T * TILE_E;
struct CompactSpec {
  int n_groups;
  int bit[COMPACT_GROUPS];
  int repeat[COMPACT_GROUPS];
};
__device__ __forceinline__ int block_scan_256(int v, int* s_scan, int& total) {  // inclusive scan over 256 threads
  const int t = threadIdx.x;
  s_scan[t] = v;
  __syncthreads();
  for (int d = 1; d < TILE_T; d <<= 1) {
    const int u = t >= d ? s_scan[t - d] : 0;
    __syncthreads();
    s_scan[t] += u;
    __syncthreads();
  }
  total = s_scan[TILE_T - 1];
  const int r = s_scan[t];
  __syncthreads();
  return r;
}
__global__ void __launch_bounds__(TILE_T) compact_count_kernel(int P, const uint8_t* __restrict__ flags, CompactSpec spec,
    int* __restrict__ tile_counts /*[n_tiles][COMPACT_GROUPS]*/) {
  __shared__ int s_scan[TILE_T];
  const int i0 = blockIdx.x * CTILE + threadIdx.x * TILE_E;
  int n[COMPACT_GROUPS] = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < TILE_E; ++e) {
    const int f = i0 + e < P ? flags[i0 + e] : 0;
#pragma unroll
    for (int g = 0; g < COMPACT_GROUPS; ++g) n[g] += (g < spec.n_groups && (f & spec.bit[g])) ? 1 : 0;
  }
#pragma unroll
  for (int g = 0; g < COMPACT_GROUPS; ++g) {
    int total;
    block_scan_256(n[g], s_scan, total);
    if (threadIdx.x == 0) tile_counts[blockIdx.x * COMPACT_GROUPS + g] = total;
  }
}
// exclusive scan of the tile counts per group (in place) + the group sizes
__global__ void __launch_bounds__(TILE_T) compact_scan_kernel(int n_tiles, CompactSpec spec, int* __restrict__ tile_counts,
    int32_t* __restrict__ counts) {
  __shared__ int s_scan[TILE_T];
  for (int g = 0; g < spec.n_groups; ++g) {
    int carry = 0;
    for (int base = 0; base < n_tiles; base += TILE_T) {
      const int i = base + threadIdx.x;
      const int v = i < n_tiles ? tile_counts[i * COMPACT_GROUPS + g] : 0;
      int total;
      const int inc = block_scan_256(v, s_scan, total);
      if (i < n_tiles) tile_counts[i * COMPACT_GROUPS + g] = carry + inc - v;
      carry += total;
    }
    if (threadIdx.x == 0) counts[g] = carry;
  }
}
__global__ void __launch_bounds__(TILE_T) compact_scatter_kernel(int P, const uint8_t* __restrict__ flags, CompactSpec spec,
    const int* __restrict__ tile_offsets, const int32_t* __restrict__ counts, int64_t* __restrict__ rows) {
  __shared__ int s_scan[TILE_T];
  const int i0 = blockIdx.x * CTILE + threadIdx.x * TILE_E;
  int f[TILE_E];
#pragma unroll
  for (int e = 0; e < TILE_E; ++e) f[e] = i0 + e < P ? flags[i0 + e] : 0;
  long long base = 0;  // output offset of the current group's first block
  for (int g = 0; g < spec.n_groups; ++g) {
    int n = 0;
#pragma unroll
    for (int e = 0; e < TILE_E; ++e) n += (f[e] & spec.bit[g]) ? 1 : 0;
    int total;
    const int inc = block_scan_256(n, s_scan, total);
    const long long group_n = counts[g];
    long long o = base + tile_offsets[blockIdx.x * COMPACT_GROUPS + g] + (inc - n);
#pragma unroll
    for (int e = 0; e < TILE_E; ++e)
      if (f[e] & spec.bit[g]) {
        for (int r = 0; r < spec.repeat[g]; ++r) rows[o + r * group_n] = i0 + e;
        ++o;
      }
    base += group_n * spec.repeat[g];
  }
}
// tile_ws: ceil(P / 2048) * 4 ints of scratch
int launch_compact(int P, const uint8_t* flags, const CompactSpec& spec, int64_t* rows, int32_t* counts, int* tile_ws,
    hipStream_t s) {
  const int n_tiles = (P + CTILE - 1) / CTILE;
  if (n_tiles > 0) {
    hipLaunchKernelGGL(compact_count_kernel, dim3(n_tiles), dim3(TILE_T), 0, s, P, flags, spec, tile_ws);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(TILE_T), 0, s, n_tiles, spec, tile_ws, counts);
  SKGS_CHECK_HIP(hipGetLastError());
  if (n_tiles > 0) {
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(n_tiles), dim3(TILE_T), 0, s, P, flags, spec, tile_ws, counts, rows);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

// the N samples of every split Gaussian, in place on the freshly gathered child rows (copies of their parents): a sample
// of N(mu, R diag(s)^2 R^T) for the position, scale / (0.8 N) for the scale (:599-610).  normals: standard normal draws.
__device__ __forceinline__ void quat_to_R(const float* q, float (&R)[9]) {
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);  // F.normalize
  const float x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;                       // xyzw (quaternion.py:162-172)
  R[0] = 1 - 2 * y * y - 2 * z * z, R[1] = 2 * x * y - 2 * w * z, R[2] = 2 * w * y + 2 * x * z;
  R[3] = 2 * x * y + 2 * w * z, R[4] = 1 - 2 * x * x - 2 * z * z, R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y, R[7] = 2 * w * x + 2 * y * z, R[8] = 1 - 2 * x * x - 2 * y * y;
}
__global__ void __launch_bounds__(256) split_children_kernel(int n, int N, const float* __restrict__ normals, float* __restrict__ xyz,
    float* __restrict__ log_scale, const float* __restrict__ rot) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float R[9];
  quat_to_R(rot + 4 * i, R);
  const float s[3] = {expf(log_scale[3 * i]), expf(log_scale[3 * i + 1]), expf(log_scale[3 * i + 2])};
  const float v[3] = {normals[3 * i] * s[0], normals[3 * i + 1] * s[1], normals[3 * i + 2] * s[2]};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    xyz[3 * i + a] = ((R[3 * a] * v[0] + R[3 * a + 1] * v[1]) + R[3 * a + 2] * v[2]) + xyz[3 * i + a];
    log_scale[3 * i + a] = logf(s[a] / (0.8f * (float) N));
  }
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" size_t skgs_row_tensor_bytes(void) { return sizeof(RowTensor); }

extern "C" int skgs_gather_rows(int32_t n_tensors, const void* tensors, int64_t n_out, int64_t n_keep, const int64_t* rows,
    int32_t max_row_floats, skgs_stream_t stream) {
  SKGS_REQUIRE(n_tensors >= 0 && n_out >= 0 && n_keep >= 0 && n_keep <= n_out && max_row_floats >= 1, "gather_rows: bad sizes");
  if (n_tensors == 0 || n_out == 0) return 0;
  SKGS_REQUIRE(tensors && rows, "gather_rows: NULL argument");
  const long long total = (long long) n_out * max_row_floats;
  const unsigned gx     = (unsigned) std::min<long long>((total + 1023) / 1024, 8192);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, (unsigned) n_tensors), dim3(256), 0, (hipStream_t) stream,
      reinterpret_cast<const RowTensor*>(tensors), (long long) n_out, (long long) n_keep, rows);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int skgs_densify_stats(int32_t P, const int32_t* radii, const float* grad_means2D, float grad_multiplier,
    float* xyz_gradient_accum, float* denom, float* max_radii2D, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0, "densify_stats: P must be >= 0");
  if (P == 0) return 0;
  SKGS_REQUIRE(radii && grad_means2D && xyz_gradient_accum && denom && max_radii2D, "densify_stats: NULL argument");
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t) stream, P, radii, grad_means2D,
      grad_multiplier, xyz_gradient_accum, denom, max_radii2D);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

/* Clone / split decisions and the row list of the ONE gather that realises them (densify(), gaussian_splatting.py:
 * 589-641): rows = [not split] ++ [clone] ++ [split] x N, counts (DEVICE int32[3]) = the three group sizes (read them back
 * to size the new tensors: n_keep = counts[0], n_out = counts[0] + counts[1] + N counts[2]).  rows: capacity (2 + N) P;
 * flags_ws: skgs_select_workspace_bytes(P) of scratch. */
extern "C" size_t skgs_select_workspace_bytes(int32_t P) {
  return (((size_t) std::max(P, 0) + 15) & ~(size_t) 15) + (size_t) ((std::max(P, 0) + CTILE - 1) / CTILE) * COMPACT_GROUPS * 4 + 16;
}
extern "C" int skgs_densify_select(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* log_scale,
    float max_grad, float scene_extent, int32_t N, int64_t* rows, int32_t* counts, uint8_t* flags_ws, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && N >= 1, "densify_select: bad sizes");
  SKGS_REQUIRE(counts, "densify_select: NULL counts");
  hipStream_t s = (hipStream_t) stream;
  if (P > 0) {
    SKGS_REQUIRE(xyz_gradient_accum && denom && log_scale && rows && flags_ws, "densify_select: NULL argument");
    hipLaunchKernelGGL(densify_flags_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, xyz_gradient_accum, denom, log_scale,
        max_grad, scene_extent, flags_ws);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  CompactSpec spec{3, {FLAG_KEEP, FLAG_CLONE, FLAG_SPLIT, 0}, {1, 1, N, 0}};
  return launch_compact(P, flags_ws, spec, rows, counts, reinterpret_cast<int*>(flags_ws + (((size_t) P + 15) & ~(size_t) 15)), s);
}

/* prune(): rows = the surviving Gaussians in index order, counts (DEVICE int32[1]) their number.  max_radii2D = NULL: only
 * the opacity test (the reference's max_screen_size = 0 case).  rows: capacity P. */
extern "C" int skgs_prune_select(int32_t P, const float* opacity_logit, const float* max_radii2D, const float* log_scale,
    float min_opacity, float max_screen_size, float world_size_limit, int64_t* rows, int32_t* counts, uint8_t* flags_ws,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && counts, "prune_select: bad arguments");
  hipStream_t s = (hipStream_t) stream;
  if (P > 0) {
    SKGS_REQUIRE(opacity_logit && rows && flags_ws && (!max_radii2D || log_scale), "prune_select: NULL argument");
    hipLaunchKernelGGL(prune_flags_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, opacity_logit, max_radii2D, log_scale,
        min_opacity, max_screen_size, world_size_limit, flags_ws);
    SKGS_CHECK_HIP(hipGetLastError());
  }
  CompactSpec spec{1, {FLAG_KEEP, 0, 0, 0}, {1, 0, 0, 0}};
  return launch_compact(P, flags_ws, spec, rows, counts, reinterpret_cast<int*>(flags_ws + (((size_t) P + 15) & ~(size_t) 15)), s);
}

/* The n freshly gathered children of split Gaussians (N per parent), in place: position += R(rot) (normals * exp(log_scale)),
 * log_scale = log(exp(log_scale) / (0.8 N)).  normals: [n,3] standard normal draws. */
extern "C" int skgs_split_children(int32_t n, int32_t N, const float* normals, float* xyz, float* log_scale, const float* rot,
    skgs_stream_t stream) {
  SKGS_REQUIRE(n >= 0 && N >= 1, "split_children: bad sizes");
  if (n == 0) return 0;
  SKGS_REQUIRE(normals && xyz && log_scale && rot, "split_children: NULL argument");
  hipLaunchKernelGGL(split_children_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, n, N, normals, xyz,
      log_scale, rot);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

/* ---- simple_knn: the scale initialisation of create_from_pcd (networks/gaussian_splatting.py:211-213) ----
 * The reference's op (my_ext/_C/src/other/knn.cu:113-190, upstream simple-knn): for every point the MEAN of the squared
 * distances to its three nearest other points, `(best[0] + best[1] + best[2]) / 3` with d = dx dx + dy dy + dz dz.  Upstream prunes
 * with Morton-sorted boxes; the result is the exact three smallest distances either way, so this is a plain tiled scan: one lane
 * per query, the points streamed through LDS 256 at a time, the three best kept sorted by v_min / v_med3 (O(P^2): 1.6 ms at
 * P = 100k, and it runs once per training run -- num_init_points is 2 000 in exps/d_nerf.yaml).  Fewer than four points: the
 * missing neighbours count as FLT_MAX, as upstream. */
namespace skgs {
namespace {
__global__ void __launch_bounds__(256) simple_knn_kernel(int P, const float* __restrict__ points, float* __restrict__ out) {
  __shared__ float4 s_p[256];
  const int n  = blockIdx.x * 256 + threadIdx.x;
  const int nn = min(n, P - 1);
  const float px = points[3 * nn], py = points[3 * nn + 1], pz = points[3 * nn + 2];
  float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
  for (int base = 0; base < P; base += 256) {
    const int m = base + threadIdx.x;
    __syncthreads();
    s_p[threadIdx.x] = m < P ? make_float4(points[3 * m], points[3 * m + 1], points[3 * m + 2], 0.f) : make_float4(0, 0, 0, 0);
    __syncthreads();
    const int cnt = min(256, P - base);
    for (int i = 0; i < cnt; ++i) {
      const float4 c = s_p[i];
      const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
      float d = dx * dx + dy * dy + dz * dz;
      d = (base + i == nn) ? 3.402823466e+38f : d;  // not itself
      b2 = __builtin_amdgcn_fmed3f(b1, b2, d);      // b0 <= b1 <= b2: each slot clamps d into its interval
      b1 = __builtin_amdgcn_fmed3f(b0, b1, d);
      b0 = fminf(b0, d);
    }
  }
  if (n < P) out[n] = (b0 + b1 + b2) / 3.0f;
}
}  // namespace
}  // namespace skgs

extern "C" int skgs_simple_knn(int32_t P, const float* points, float* mean_dist2, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0, "simple_knn: P < 0");
  if (P == 0) return 0;
  SKGS_REQUIRE(points && mean_dist2, "simple_knn: NULL argument");
  hipLaunchKernelGGL(skgs::simple_knn_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t) stream, P, points, mean_dist2);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}
