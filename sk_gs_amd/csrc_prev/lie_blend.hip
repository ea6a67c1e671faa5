// lie_blend.hip -- the hot pattern of the reference's deform AS THE REFERENCE WRITES IT, for the lietorch stand-in
// (sk_gs_amd/lietorch.py):
//
//     (sk_T[indices].act(points[:, None]) * weights[..., None]).sum(dim=1)          networks/sk_gs.py:1147, :814, :1478
//     spT[self.p2sp].act(points)                                                     networks/sk_gs.py:816, :1481   (K = 1, no weights)
//
// lietorch runs this as a gather of [P,K,7] group elements, a broadcast copy of the points to [P,K,3], the `act` kernel, a
// multiply and a sum -- and the same chain backwards, ending in an index_add of P*K tangent rows into the M bones.  Here it is
// one launch per direction on the [M,7] table in LDS, with lietorch's GRADIENT CONVENTION at the boundary (this is what makes it
// a drop-in under the unmodified reference, where the chain above `sk_T` is made of lietorch ops): the gradient of a group
// element is the LEFT-TANGENT row (tau, phi) in the first 6 of its 7 slots (my_ext/_C/src/ops_3d/lie_cpu.cpp:217-236:
// dX = dq [I | -hat(X p)], dp = dq R), summed over the rows that gathered the bone (the gather's index_add).
//
// Arithmetic: the SE3 constructor normalises the quaternion (lie.h:45-47), act is p + w uv + q x uv + t with uv = 2 q x p
// (lie.h:59-64,246) -- `se3_act` of deform_lane.h, shared with the fused skeleton-stage kernels, fp contract off.
#include "deform_lane.h"
#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

constexpr int BLEND_THREADS  = 256;
constexpr int BLEND_LDS_MAXM = 1024;  // (7 + 6) * 1024 * 4 B = 52 KB: table + the workgroup's gradient rows
constexpr int BLEND_MAX_WG   = 512;   // persistent workgroups of the backward (one partial [M,6] each)

struct BlendArgs {
  int P, K, M;
  const float* T;
  const int64_t* indices;
  const float* points;
  const float* weights;  // NULL: every weight is 1 (and there is no weight gradient)
  float* out;
  const float* g_out;
  float* g_T;
  float* g_weights;
  float* g_points;
  float* partials;  // [gridDim.x][M][6]
};

__device__ __forceinline__ void stage_table(const BlendArgs& a, float* s_T) {
  for (int j = threadIdx.x; j < a.M; j += BLEND_THREADS) {
    const float q0 = a.T[7 * j + 3], q1 = a.T[7 * j + 4], q2 = a.T[7 * j + 5], q3 = a.T[7 * j + 6];
    const float n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    float* b = s_T + 7 * j;
    b[0] = q0 / n, b[1] = q1 / n, b[2] = q2 / n, b[3] = q3 / n;
    b[4] = a.T[7 * j], b[5] = a.T[7 * j + 1], b[6] = a.T[7 * j + 2];
  }
}
// a bone beyond the LDS table's capacity: normalised on the fly from global memory
__device__ __forceinline__ void load_row(const BlendArgs& a, int j, float* b) {
  const float q0 = a.T[7 * j + 3], q1 = a.T[7 * j + 4], q2 = a.T[7 * j + 5], q3 = a.T[7 * j + 6];
  const float n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  b[0] = q0 / n, b[1] = q1 / n, b[2] = q2 / n, b[3] = q3 / n;
  b[4] = a.T[7 * j], b[5] = a.T[7 * j + 1], b[6] = a.T[7 * j + 2];
}

// torch's index semantics for the int64 neighbour indices: a negative index counts from the end (j += M); one outside
// [-M, M) -- torch raises a device assertion there -- addresses nothing: the entry contributes nothing, forward and backward, instead of
// reading or adding through a wild LDS / global address
__device__ __forceinline__ int wrap_row(long long v, int M) {
  if (v < 0) v += M;
  return (v < 0 || v >= M) ? -1 : (int) v;
}

template <bool IN_LDS>
__global__ void __launch_bounds__(BLEND_THREADS) se3_blend_forward_kernel(const BlendArgs a) {
  extern __shared__ float s_mem[];
  if (IN_LDS) {
    stage_table(a, s_mem);
    __syncthreads();
  }
  for (int n = blockIdx.x * BLEND_THREADS + threadIdx.x; n < a.P; n += gridDim.x * BLEND_THREADS) {
    const float p[3] = {a.points[3 * n], a.points[3 * n + 1], a.points[3 * n + 2]};
    float s[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; ++k) {
      const int j   = wrap_row(a.indices[(size_t) n * a.K + k], a.M);
      if (j < 0) continue;
      float row[7];
      const float* b = s_mem + 7 * j;
      if (!IN_LDS) load_row(a, j, row), b = row;
      float y[3];
      se3_act(b, p, y);
      if (a.weights) {
        const float w = a.weights[(size_t) n * a.K + k];
        s[0] += y[0] * w, s[1] += y[1] * w, s[2] += y[2] * w;
      } else {
        s[0] += y[0], s[1] += y[1], s[2] += y[2];
      }
    }
    a.out[3 * n] = s[0], a.out[3 * n + 1] = s[1], a.out[3 * n + 2] = s[2];
  }
}

// IN_LDS: the bone gradient rows are summed in LDS per workgroup (ds_add_f32), each workgroup writes ONE partial [M,6], a second
// launch adds the partials in workgroup order.  Otherwise (M > 1024): global atomics into the zero-filled g_T.
template <bool IN_LDS>
__global__ void __launch_bounds__(BLEND_THREADS) se3_blend_backward_kernel(const BlendArgs a) {
  extern __shared__ float s_mem[];
  float* s_T   = s_mem;
  float* s_acc = s_mem + 7 * a.M;
  if (IN_LDS) {
    stage_table(a, s_T);
    for (int i = threadIdx.x; i < 6 * a.M; i += BLEND_THREADS) s_acc[i] = 0.f;
    __syncthreads();
  }
  for (int n = blockIdx.x * BLEND_THREADS + threadIdx.x; n < a.P; n += gridDim.x * BLEND_THREADS) {
    const float p[3] = {a.points[3 * n], a.points[3 * n + 1], a.points[3 * n + 2]};
    const float g[3] = {a.g_out[3 * n], a.g_out[3 * n + 1], a.g_out[3 * n + 2]};
    float gp[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; ++k) {
      const int j = wrap_row(a.indices[(size_t) n * a.K + k], a.M);
      if (j < 0) {
        if (a.g_weights) a.g_weights[(size_t) n * a.K + k] = 0.f;
        continue;
      }
      float row[7];
      const float* b = s_T + 7 * j;
      if (!IN_LDS) load_row(a, j, row), b = row;
      float y[3];
      se3_act(b, p, y);
      const float w = a.weights ? a.weights[(size_t) n * a.K + k] : 1.f;
      if (a.g_weights) a.g_weights[(size_t) n * a.K + k] = g[0] * y[0] + g[1] * y[1] + g[2] * y[2];
      const float c[3] = {g[0] * w, g[1] * w, g[2] * w};  // cotangent of y = X p
      // dX = c [I | -hat(y)]  (lie_cpu.cpp:234, act_jacobian lie.h:387-393): tau = c, phi = y x c
      const float t6[6] = {c[0], c[1], c[2], y[1] * c[2] - y[2] * c[1], y[2] * c[0] - y[0] * c[2], y[0] * c[1] - y[1] * c[0]};
      if (IN_LDS) {
#pragma unroll
        for (int e = 0; e < 6; ++e) atomicAdd(s_acc + 6 * j + e, t6[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 6; ++e) atomicAdd(a.g_T + 7 * j + e, t6[e]);
      }
      if (a.g_points) {
        // dp = c R  (lie_cpu.cpp:233) = R^T c: the rotation by the conjugate quaternion
        const float qc[4] = {-b[0], -b[1], -b[2], b[3]};
        float uv[3] = {qc[1] * c[2] - qc[2] * c[1], qc[2] * c[0] - qc[0] * c[2], qc[0] * c[1] - qc[1] * c[0]};
        uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
        gp[0] += c[0] + qc[3] * uv[0] + (qc[1] * uv[2] - qc[2] * uv[1]);
        gp[1] += c[1] + qc[3] * uv[1] + (qc[2] * uv[0] - qc[0] * uv[2]);
        gp[2] += c[2] + qc[3] * uv[2] + (qc[0] * uv[1] - qc[1] * uv[0]);
      }
    }
    if (a.g_points) a.g_points[3 * n] = gp[0], a.g_points[3 * n + 1] = gp[1], a.g_points[3 * n + 2] = gp[2];
  }
  if (IN_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * a.M; i += BLEND_THREADS) a.partials[(size_t) blockIdx.x * 6 * a.M + i] = s_acc[i];
  }
}

// one WAVE per output element: its lanes take the partials 64 apart, then a fixed-order butterfly (G is up to 512 partials: a lane
// summing them alone is a chain of 512 dependent loads -- 94 us for 140 elements)
__device__ __forceinline__ float wave_sum_partials(const float* __restrict__ partials, size_t stride, size_t offset, int G) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int g = lane; g < G; g += 64) s += partials[(size_t) g * stride + offset];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}
__global__ void __launch_bounds__(BLEND_THREADS) se3_blend_reduce_kernel(int M, int G, const float* __restrict__ partials,
    float* __restrict__ g_T) {
  const int i = (blockIdx.x * BLEND_THREADS + threadIdx.x) >> 6;  // (bone, slot) over [M,7]
  if (i >= 7 * M) return;
  const int j = i / 7, e = i % 7;
  const float s = e < 6 ? wave_sum_partials(partials, (size_t) 6 * M, (size_t) 6 * j + e, G) : 0.f;
  if ((threadIdx.x & 63) == 0) g_T[i] = s;  // slot 6: lietorch's gradient buffers are 7 wide with the tangent in the first 6 (lie_cpu.cpp:489)
}

// out[idx[r], :] += g[r, :] over R rows of C floats: the backward of the reference's `table[indices]` gathers of per-bone rows
// (sk_gs.py:1148-1149 `sk_d_rot[indices]`, `sk_d_scale[indices]`, :760-763 `kernel_radius[indices]`): torch's index backward sorts
// the R = P * K indices and walks each bone's ~P * K / M duplicates serially -- 7.5 ms per gather at 100k x 5 into 20 rows on this
// GPU.  Same scheme as the blend backward: LDS rows per workgroup, one partial per workgroup, a second launch adds them in order.
template <bool IN_LDS>
__global__ void __launch_bounds__(BLEND_THREADS) index_add_rows_kernel(long long R, int C, int M, const int64_t* __restrict__ idx,
    const float* __restrict__ g, float* __restrict__ out, float* __restrict__ partials) {
  extern __shared__ float s_acc[];
  if (IN_LDS) {
    for (int i = threadIdx.x; i < M * C; i += BLEND_THREADS) s_acc[i] = 0.f;
    __syncthreads();
  }
  const long long n = R * C;
  for (long long e = (long long) blockIdx.x * BLEND_THREADS + threadIdx.x; e < n; e += (long long) gridDim.x * BLEND_THREADS) {
    const long long r = e / C;
    const int c = (int) (e - r * C);
    const int j = wrap_row(idx[r], M);
    if (j < 0) continue;
    if (IN_LDS)
      atomicAdd(s_acc + j * C + c, g[e]);
    else
      atomicAdd(out + (size_t) j * C + c, g[e]);
  }
  if (IN_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < M * C; i += BLEND_THREADS) partials[(size_t) blockIdx.x * M * C + i] = s_acc[i];
  }
}
__global__ void __launch_bounds__(BLEND_THREADS) index_add_reduce_kernel(int n, int G, const float* __restrict__ partials, float* __restrict__ out) {
  const int i = (blockIdx.x * BLEND_THREADS + threadIdx.x) >> 6;
  if (i >= n) return;
  const float s = wave_sum_partials(partials, (size_t) n, (size_t) i, G);
  if ((threadIdx.x & 63) == 0) out[i] = s;
}
constexpr int INDEX_ADD_LDS_FLOATS = 12 * 1024;  // 48 KB of rows per workgroup
inline int index_add_grid(long long n) { return (int) std::max<long long>(1, std::min<long long>((n + BLEND_THREADS - 1) / BLEND_THREADS, BLEND_MAX_WG)); }

inline int blend_grid(int P) { return std::max(1, std::min((P + BLEND_THREADS - 1) / BLEND_THREADS, BLEND_MAX_WG)); }

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

int skgs_se3_blend_forward(int32_t P, int32_t K, int32_t M, const float* T, const int64_t* indices, const float* points,
    const float* weights, float* out, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && K >= 1 && M >= 1, "se3_blend: bad sizes");
  if (P == 0) return 0;
  SKGS_REQUIRE(T && indices && points && out, "se3_blend: NULL argument");
  BlendArgs a{};
  a.P = P, a.K = K, a.M = M, a.T = T, a.indices = indices, a.points = points, a.weights = weights, a.out = out;
  const int grid = std::max(1, std::min((P + BLEND_THREADS - 1) / BLEND_THREADS, 4096));
  if (M <= BLEND_LDS_MAXM)
    hipLaunchKernelGGL(se3_blend_forward_kernel<true>, dim3(grid), dim3(BLEND_THREADS), (size_t) M * 7 * 4, (hipStream_t) stream, a);
  else
    hipLaunchKernelGGL(se3_blend_forward_kernel<false>, dim3(grid), dim3(BLEND_THREADS), 0, (hipStream_t) stream, a);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t skgs_se3_blend_backward_workspace_bytes(int32_t P, int32_t M) {
  if (M > BLEND_LDS_MAXM || P <= 0) return 16;
  return (size_t) blend_grid(P) * 6 * (size_t) M * 4 + 16;
}

int skgs_se3_blend_backward(int32_t P, int32_t K, int32_t M, const float* T, const int64_t* indices, const float* points,
    const float* weights, const float* g_out, float* g_T, float* g_weights, float* g_points, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && K >= 1 && M >= 1, "se3_blend: bad sizes");
  SKGS_REQUIRE(g_T, "se3_blend backward: NULL g_T");
  SKGS_REQUIRE(weights || !g_weights, "se3_blend backward: a weight gradient without weights");
  if (P == 0) {
    SKGS_CHECK_HIP(hipMemsetAsync(g_T, 0, (size_t) M * 7 * 4, (hipStream_t) stream));
    return 0;
  }
  SKGS_REQUIRE(T && indices && points && g_out, "se3_blend backward: NULL argument");
  BlendArgs a{};
  a.P = P, a.K = K, a.M = M, a.T = T, a.indices = indices, a.points = points, a.weights = weights, a.g_out = g_out;
  a.g_T = g_T, a.g_weights = g_weights, a.g_points = g_points;
  const int grid = blend_grid(P);
  if (M <= BLEND_LDS_MAXM) {
    SKGS_REQUIRE(workspace && workspace_bytes >= skgs_se3_blend_backward_workspace_bytes(P, M), "se3_blend backward: workspace too small");
    a.partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(se3_blend_backward_kernel<true>, dim3(grid), dim3(BLEND_THREADS), (size_t) M * 13 * 4, (hipStream_t) stream, a);
    SKGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(se3_blend_reduce_kernel, dim3((7 * M * 64 + BLEND_THREADS - 1) / BLEND_THREADS), dim3(BLEND_THREADS), 0,
        (hipStream_t) stream, M, grid, a.partials, g_T);
  } else {
    SKGS_CHECK_HIP(hipMemsetAsync(g_T, 0, (size_t) M * 7 * 4, (hipStream_t) stream));
    hipLaunchKernelGGL(se3_blend_backward_kernel<false>, dim3(grid), dim3(BLEND_THREADS), 0, (hipStream_t) stream, a);
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t skgs_index_add_rows_workspace_bytes(int64_t R, int32_t C, int32_t M) {
  if (R <= 0 || (long long) M * C > INDEX_ADD_LDS_FLOATS) return 16;
  return (size_t) index_add_grid(R * C) * (size_t) M * C * 4 + 16;
}

int skgs_index_add_rows(int64_t R, int32_t C, int32_t M, const int64_t* indices, const float* rows, float* out, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream) {
  SKGS_REQUIRE(R >= 0 && C >= 1 && M >= 1 && out, "index_add_rows: bad arguments");
  if (R == 0) {
    SKGS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t) M * C * 4, (hipStream_t) stream));
    return 0;
  }
  SKGS_REQUIRE(indices && rows, "index_add_rows: NULL argument");
  const int grid = index_add_grid(R * C);
  if ((long long) M * C <= INDEX_ADD_LDS_FLOATS) {
    SKGS_REQUIRE(workspace && workspace_bytes >= skgs_index_add_rows_workspace_bytes(R, C, M), "index_add_rows: workspace too small");
    float* partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(index_add_rows_kernel<true>, dim3(grid), dim3(BLEND_THREADS), (size_t) M * C * 4, (hipStream_t) stream, (long long) R, C, M,
        indices, rows, out, partials);
    SKGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(index_add_reduce_kernel, dim3((M * C * 64 + BLEND_THREADS - 1) / BLEND_THREADS), dim3(BLEND_THREADS), 0, (hipStream_t) stream,
        M * C, grid, partials, out);
  } else {
    SKGS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t) M * C * 4, (hipStream_t) stream));
    hipLaunchKernelGGL(index_add_rows_kernel<false>, dim3(grid), dim3(BLEND_THREADS), 0, (hipStream_t) stream, (long long) R, C, M, indices, rows, out,
        (float*) nullptr);
  }
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
