// preprocess.hip -- per-Gaussian kernels of the rasterizer (gfx950).
//
//   preprocess_forward_kernel : cull, project, Sigma3D -> Sigma2D -> conic, screen radius, tile rect, SH -> RGB,
//                               per-tile instance counts.            (reference: gaussian_preprocess.cu:99-168,
//                               gaussian_preprocess_colmap.cu:155-224, gaussian_rasterizer_forward.cu:97-137)
//   preprocess_backward_kernel: conic -> Sigma2D -> Sigma3D/mean, projection, SH and Sigma3D -> scale/rotation
//                               gradients in ONE pass (reference runs computeCov2DCUDA + preprocessCUDA_backward:
//                               gaussian_preprocess.cu:183-400, gaussian_preprocess_colmap.cu:240-481,
//                               gaussian_rasterizer_backwrad.cu:26-127)
//
// HBM-bound streaming kernels, one lane per Gaussian.  The arithmetic is written in the reference's evaluation
// order and compiled without FMA contraction so that every discrete decision (cull, radius = ceil(..), tile
// rectangle, SH clamp) is bit-identical to the CPU oracle; these kernels move ~0.3 KB per Gaussian, the extra
// VALU work is free.
#include <algorithm>

#include "skgs_common.h"
#include "deform_lane.h"

#pragma clang fp contract(off)

namespace skgs {
namespace {

__device__ const float SH_C0   = 0.28209479177387814f;
__device__ const float SH_C1   = 0.4886025119029199f;
__device__ const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
    -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
    0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct Cam {
  float view[16];
  float proj[16];
  float campos[3];
};

// A 3x3 held as m[c][r] (column-major) with the product order  r[c][w] = a[0][w]*b[c][0] + a[1][w]*b[c][1] + a[2][w]*b[c][2]
struct M3 {
  float m[3][3];
};
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int w = 0; w < 3; ++w) r.m[c][w] = a.m[0][w] * b.m[c][0] + a.m[1][w] * b.m[c][1] + a.m[2][w] * b.m[c][2];
  return r;
}
__device__ __forceinline__ M3 m3_t(const M3& a) {
  M3 r;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int w = 0; w < 3; ++w) r.m[c][w] = a.m[w][c];
  return r;
}
__device__ __forceinline__ M3 m3_sym(const float* c6) {
  M3 v;
  v.m[0][0] = c6[0], v.m[0][1] = c6[1], v.m[0][2] = c6[2];
  v.m[1][0] = c6[1], v.m[1][1] = c6[3], v.m[1][2] = c6[4];
  v.m[2][0] = c6[2], v.m[2][1] = c6[4], v.m[2][2] = c6[5];
  return v;
}

// ---------------------------------------------------------------------------------------------- colmap = 1
__device__ __forceinline__ void xf3_cm(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xf4_cm(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
__device__ __forceinline__ M3 rot_cm(const float* q) {
  const float x = q[0], y = q[1], z = q[2], r = q[3];
  M3 R;
  R.m[0][0] = 1.f - 2.f * (y * y + z * z), R.m[0][1] = 2.f * (x * y - r * z), R.m[0][2] = 2.f * (x * z + r * y);
  R.m[1][0] = 2.f * (x * y + r * z), R.m[1][1] = 1.f - 2.f * (x * x + z * z), R.m[1][2] = 2.f * (y * z - r * x);
  R.m[2][0] = 2.f * (x * z - r * y), R.m[2][1] = 2.f * (y * z + r * x), R.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return R;
}
__device__ __forceinline__ M3 scale_rot_cm(const float* s /*already * mod*/, const M3& R) {
  M3 S;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int w = 0; w < 3; ++w) S.m[c][w] = (c == w) ? s[c] : 0.f;
  return m3_mul(S, R);
}
__device__ __forceinline__ void cov3d_cm(const float* scale, float mod, const float* q, float* c6) {
  const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  M3 Mm            = scale_rot_cm(s, rot_cm(q));
  M3 Sg            = m3_mul(m3_t(Mm), Mm);
  c6[0] = Sg.m[0][0], c6[1] = Sg.m[0][1], c6[2] = Sg.m[0][2], c6[3] = Sg.m[1][1], c6[4] = Sg.m[1][2], c6[5] = Sg.m[2][2];
}
// T = W*J and the clamped view-space point; shared by forward and backward
struct ProjCM {
  M3 T, W, V;
  float t[3];
  float xm, ym;
};
__device__ __forceinline__ ProjCM proj_cm(const float* mean, float fx, float fy, float tfx, float tfy, const float* c6,
    const float* vm) {
  ProjCM o;
  xf3_cm(mean, vm, o.t);
  const float limx = 1.3f * tfx, limy = 1.3f * tfy;
  const float txtz = o.t[0] / o.t[2], tytz = o.t[1] / o.t[2];
  o.t[0]           = fminf(limx, fmaxf(-limx, txtz)) * o.t[2];
  o.t[1]           = fminf(limy, fmaxf(-limy, tytz)) * o.t[2];
  o.xm             = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  o.ym             = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  M3 J;
  J.m[0][0] = fx / o.t[2], J.m[0][1] = 0.f, J.m[0][2] = -(fx * o.t[0]) / (o.t[2] * o.t[2]);
  J.m[1][0] = 0.f, J.m[1][1] = fy / o.t[2], J.m[1][2] = -(fy * o.t[1]) / (o.t[2] * o.t[2]);
  J.m[2][0] = 0.f, J.m[2][1] = 0.f, J.m[2][2] = 0.f;
  o.W.m[0][0] = vm[0], o.W.m[0][1] = vm[4], o.W.m[0][2] = vm[8];
  o.W.m[1][0] = vm[1], o.W.m[1][1] = vm[5], o.W.m[1][2] = vm[9];
  o.W.m[2][0] = vm[2], o.W.m[2][1] = vm[6], o.W.m[2][2] = vm[10];
  o.T = m3_mul(o.W, J);
  o.V = m3_sym(c6);
  return o;
}

// ---------------------------------------------------------------------------------------------- colmap = 0
__device__ __forceinline__ void xf3_rm(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
}
__device__ __forceinline__ void xf4_rm(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
  o[3] = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
}
__device__ __forceinline__ void q2R_rm(const float* q, float* R) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - z * w), R[2] = 2 * (y * w + x * z);
  R[3] = 2 * (x * y + z * w), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w), R[7] = 2 * (x * w + y * z), R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void mm_rm(const float* A, const float* B, float* C) {  // C += A*B, k innermost
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) C[i * 3 + j] += A[i * 3 + k] * B[k * 3 + j];
}
__device__ __forceinline__ void mm_tn_rm(const float* At, const float* B, float* C) {  // C += At^T * B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) C[i * 3 + j] += At[k * 3 + i] * B[k * 3 + j];
}
__device__ __forceinline__ void cov3d_rm(const float* s, const float* q, float* c6) {
  float R[9];
  q2R_rm(q, R);
  const float sx2 = s[0] * s[0], sy2 = s[1] * s[1], sz2 = s[2] * s[2];
  c6[0] = R[0] * R[0] * sx2 + R[1] * R[1] * sy2 + R[2] * R[2] * sz2;
  c6[1] = R[0] * R[3] * sx2 + R[1] * R[4] * sy2 + R[2] * R[5] * sz2;
  c6[2] = R[0] * R[6] * sx2 + R[1] * R[7] * sy2 + R[2] * R[8] * sz2;
  c6[3] = R[3] * R[3] * sx2 + R[4] * R[4] * sy2 + R[5] * R[5] * sz2;
  c6[4] = R[3] * R[6] * sx2 + R[4] * R[7] * sy2 + R[5] * R[8] * sz2;
  c6[5] = R[6] * R[6] * sx2 + R[7] * R[7] * sy2 + R[8] * R[8] * sz2;
}
struct ProjRM {
  float T[9], W[9];
  float t[3];
  float xm, ym;
};
__device__ __forceinline__ ProjRM proj_rm(const float* mean, float fx, float fy, float tfx, float tfy, const float* vm) {
  ProjRM o;
  xf3_rm(mean, vm, o.t);
  const float limx = 1.3f * tfx, limy = 1.3f * tfy;
  const float txtz = o.t[0] / o.t[2], tytz = o.t[1] / o.t[2];
  o.t[0]           = fminf(fmaxf(txtz, -limx), limx) * o.t[2];
  o.t[1]           = fminf(fmaxf(tytz, -limy), limy) * o.t[2];
  o.xm             = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  o.ym             = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float J[9] = {fx / o.t[2], 0.f, -(fx * o.t[0]) / (o.t[2] * o.t[2]), 0.f, fy / o.t[2],
      -(fy * o.t[1]) / (o.t[2] * o.t[2]), 0.f, 0.f, 0.f};
  o.W[0] = vm[0], o.W[1] = vm[1], o.W[2] = vm[2], o.W[3] = vm[4], o.W[4] = vm[5], o.W[5] = vm[6];
  o.W[6] = vm[8], o.W[7] = vm[9], o.W[8] = vm[10];
#pragma unroll
  for (int i = 0; i < 9; ++i) o.T[i] = 0.f;
  mm_rm(o.W, J, o.T);
  return o;
}
__device__ __forceinline__ void cov2d_rm(const ProjRM& pr, const float* c6, float* cov) {
  const float V[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
  float tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  mm_tn_rm(pr.T, V, tmp);
  mm_rm(tmp, pr.T, c);
  cov[0] = c[0] + 0.3f, cov[1] = c[1], cov[2] = c[4] + 0.3f;
}

// ---------------------------------------------------------------------------------------------- shared
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float) ((((double) v + 1.0) * S - 1.0) * 0.5); }

// SH basis * coefficients for one Gaussian.  Coefficient 0 is read through `dc`, coefficients 1.. through `sh`
// (indexed from coefficient 0): one [M][3] row has dc == sh; split DC / rest storage (skgs_raster_inputs::sh_rest)
// passes two rows, the second biased by -3 floats.
__device__ __forceinline__ void sh_to_rgb(int deg, const float* mean, const float* campos, const float* dc, const float* sh, float* rgb,
    uint32_t* clamp_bits) {
  float d[3]      = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
  const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float x = d[0] / len, y = d[1] / len, z = d[2] / len;
  uint32_t bits = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float r = SH_C0 * dc[c];
    if (deg > 0) {
      r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
            SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
        if (deg > 2) {
          r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
              SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
        }
      }
    }
    r += 0.5f;
    if (r < 0.f) bits |= (1u << c);
    rgb[c] = fmaxf(r, 0.0f);
  }
  *clamp_bits = bits;
}

// SH rows of a workgroup, staged through LDS.  A lane needs its own row of RL floats (45 for the split "rest" storage,
// 48 for one [16][3] row): read directly, every load instruction of a wave touches 64 different rows and the L1 cannot
// hold the workgroup's span, so each 64-B line is fetched from L2 many times.  Here the waves copy whole rows with
// consecutive lanes on consecutive floats (the span of a workgroup is contiguous in memory); rows get an odd pitch so
// that the per-lane reads afterwards are bank-conflict free.
constexpr int PRE_THREADS     = 128;
// the backward in workgroups of 256 (50 KB of staged SH rows each, three per CU): 20.9 us against 22.1 at 100k Gaussians; the
// forward is slower that way and with 64 (A/B on one box, 300 steps each)
constexpr int PRE_BWD_THREADS = 256;
__device__ __forceinline__ int sh_pitch(int RL) { return RL | 1; }
// Both copies keep all of a thread's global accesses in flight at once (up to STAGE_V float4 per thread): a loop of
// load -> LDS store per row serialised on the load latency and was slower than no staging at all.
constexpr int STAGE_V = 12;  // NT rows x 48 floats / 4 / NT threads
__device__ __forceinline__ int stage_lds_index(int e, int RL, int pitch) {  // float index e of the span -> LDS index
  if (pitch == RL) return e;
  const int r = e / RL;
  return r * pitch + (e - r * RL);
}
template <int NT = PRE_THREADS, int SITE = 31 /* StreamSite of the loads (31: plain) */>
__device__ __forceinline__ void stage_rows_in(float* s_dst, const float* __restrict__ src, int nrows, int RL) {
  const int pitch = sh_pitch(RL), n = nrows * RL;
  const bool vec  = (RL % 4 == 0 || pitch == RL) && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  const int n4    = vec ? n >> 2 : 0;
  float4 v[STAGE_V];
#pragma unroll
  for (int k = 0; k < STAGE_V; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < n4) v[k] = stream_load4<SITE>(src + 4 * (size_t) i);
  }
#pragma unroll
  for (int k = 0; k < STAGE_V; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < n4) {
      if (pitch == RL) {
        s_dst[4 * i] = v[k].x, s_dst[4 * i + 1] = v[k].y, s_dst[4 * i + 2] = v[k].z, s_dst[4 * i + 3] = v[k].w;
      } else {  // RL % 4 == 0: the four floats stay in one row
        const int o = stage_lds_index(4 * i, RL, pitch);
        s_dst[o] = v[k].x, s_dst[o + 1] = v[k].y, s_dst[o + 2] = v[k].z, s_dst[o + 3] = v[k].w;
      }
    }
  }
  for (int e = 4 * n4 + threadIdx.x; e < n; e += NT) s_dst[stage_lds_index(e, RL, pitch)] = src[e];
}
template <int NT = PRE_THREADS, int SITE = 31 /* StreamSite of the stores (31: plain) */>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ dst, const float* s_src, int nrows, int RL) {
  const int pitch = sh_pitch(RL), n = nrows * RL;
  const bool vec  = (RL % 4 == 0 || pitch == RL) && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  const int n4    = vec ? n >> 2 : 0;
#pragma unroll
  for (int k = 0; k < STAGE_V; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < n4) {
      const int o = pitch == RL ? 4 * i : stage_lds_index(4 * i, RL, pitch);
      stream_store4<SITE>(dst + 4 * (size_t) i, make_float4(s_src[o], s_src[o + 1], s_src[o + 2], s_src[o + 3]));
    }
  }
  for (int e = 4 * n4 + threadIdx.x; e < n; e += NT) stream_store<SITE>(dst + e, s_src[stage_lds_index(e, RL, pitch)]);
}

// The skeleton stage's deform in front of this pass (template DK > 0: the capacity of the per-lane top-K list): the lane first
// computes its Gaussian's mean / scale / rotation / opacity -- K nearest bones, softmax weights, skinning, activations,
// knn_deform_forward_kernel's arithmetic through the same deform_lane.h functions -- writes them (and the weights / indices) for
// the backward, and projects them from registers: one launch and one round trip through HBM less per step.
// DK == -1: the skinning alone -- weights / indices are INPUTS (joints == NULL in the public job: the superpoint stage's search has
// produced them), any number of bones, their rows gathered from global memory (deform_forward_kernel<false>'s arithmetic).
struct KnnDeformJob {
  int M, K, lds_offset /* floats: where the deform's tables start in the dynamic LDS (behind the SH rows) */;
  int largest;         /* (DK == -1) warp_method `largest`: the position follows the bone of the largest weight alone */
  const float *points, *joints, *sp_W, *bone_T, *bone_drot, *bone_dscale, *xyz, *log_scale, *rot, *opacity_logit;
  int64_t* out_idx;
  float *out_weights, *means, *scales, *rotations, *opacity;
};

template <bool COLMAP, int DK>
__global__ void __launch_bounds__(PRE_THREADS) preprocess_forward_kernel(int P, int D, int M, const float* __restrict__ means3D,
    const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations,
    const float* __restrict__ opacities, const float* __restrict__ shs, const float* __restrict__ shs_rest,
    const float* __restrict__ cov3D_precomp,
    const float* __restrict__ colors_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
    int gx, int gy, int32_t* __restrict__ radii, float4* __restrict__ recs, uint32_t* __restrict__ tile_counts,
    GeomHeader* hdr_bucket /* bucket layout: tile_counts are the per-tile cursors and the status words start here */,
    const float* __restrict__ tanfov_dev /* NULL, or {tanfovx, tanfovy} read here instead of the launch arguments */,
    const int32_t* __restrict__ live /* NULL, or the live Gaussian count (<= P, the capacity the grid was sized for) */,
    KnnDeformJob dj) {
  // the number of Gaussians as a device word (one captured graph survives densification): rows [live, P) of the capacity
  // get an all-zero record and radius 0 -- what a culled Gaussian gets -- so nothing downstream needs to know
  const int P_cap = P;
  if (live) P = min(P, live[0]);
  if (tanfov_dev) {  // same expressions as the host launcher: identical bits
    tan_fovx = tanfov_dev[0], tan_fovy = tanfov_dev[1];
    focal_x = W / (2.0f * tan_fovx), focal_y = H / (2.0f * tan_fovy);
  }
  // Every per-Gaussian input of the lane is requested HERE, before the camera and the SH rows are staged: the kernel runs at
  // 1.5 waves per SIMD and 64 % of its wave time was spent in s_waitcnt (tools/pmc_kernel.sh) on a chain of four dependent
  // round trips -- camera, SH staging, mean, then scale / rotation / opacity behind the cull test.  Now they are one.  (Rows up
  // to the capacity exist; a lane behind it reads row 0 and uses nothing.)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ld  = idx < P_cap ? idx : 0;
  float pf_p[3] = {0.f, 0.f, 0.f}, pf_op = 0.f;
  float pf_s[3] = {0.f, 0.f, 0.f}, pf_c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pf_col[3] = {0.f, 0.f, 0.f};
  float4 pf_q = make_float4(0.f, 0.f, 0.f, 1.f);
  // (DK > 0) the deform's per-Gaussian inputs instead: the point the bones are searched from, and the four raw parameters
  float dj_p[3] = {0.f, 0.f, 0.f}, dj_x[3] = {0.f, 0.f, 0.f}, dj_ls[3] = {0.f, 0.f, 0.f}, dj_ol = 0.f;
  float4 dj_r4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int sk_j[PREF_K];  // (DK == -1) the first PREF_K (index, weight) pairs of the lane
  float sk_w[PREF_K];
  if constexpr (DK != 0) {
    if constexpr (DK == -1) {
#pragma unroll
      for (int q = 0; q < PREF_K; ++q) {
        sk_j[q] = q < dj.K ? (int) dj.out_idx[(size_t) ld * dj.K + q] : 0;
        sk_w[q] = q < dj.K ? dj.out_weights[(size_t) ld * dj.K + q] : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dj_p[c] = dj.points[3 * ld + c], dj_x[c] = dj.xyz[3 * ld + c], dj_ls[c] = dj.log_scale[3 * ld + c];
    dj_r4 = reinterpret_cast<const float4*>(dj.rot)[ld], dj_ol = dj.opacity_logit[ld];
  } else {
    pf_p[0] = means3D[3 * ld], pf_p[1] = means3D[3 * ld + 1], pf_p[2] = means3D[3 * ld + 2];
    pf_op   = opacities[ld];
    if (cov3D_precomp != nullptr) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pf_c6[i] = cov3D_precomp[6 * ld + i];
    } else {
      pf_s[0] = scales[3 * ld], pf_s[1] = scales[3 * ld + 1], pf_s[2] = scales[3 * ld + 2];
      pf_q    = reinterpret_cast<const float4*>(rotations)[ld];
    }
  }
  if (colors_precomp != nullptr)
    pf_col[0] = colors_precomp[3 * ld], pf_col[1] = colors_precomp[3 * ld + 1], pf_col[2] = colors_precomp[3 * ld + 2];
  __shared__ Cam cam;  // (its three loads ride in the same round trip; stored to LDS behind the SH rows, ONE barrier for both)
  const float cam_v = threadIdx.x < 16 ? viewmatrix[threadIdx.x] : 0.f, cam_p = threadIdx.x < 16 ? projmatrix[threadIdx.x] : 0.f;
  const float cam_c = threadIdx.x < 3 ? campos[threadIdx.x] : 0.f;
  // the per-tile counters of the next kernel (binning.hip: count_tiles) start from zero: cleared here, not by a fill launch
  for (int t = idx; t < gx * gy; t += gridDim.x * blockDim.x) tile_counts[t] = 0u;
  if (hdr_bucket && idx == 0) {  // no scan kernel in the bucket layout: R and the longest list are not computed
    hdr_bucket->num_rendered = -1, hdr_bucket->max_tile_count = -1, hdr_bucket->overflow = 0, hdr_bucket->big_tiles = 0;
  }
  if ((int) (blockIdx.x * blockDim.x) >= P) {  // a workgroup of the capacity's slack rows: culled-Gaussian outputs, no staging
    if (idx < P_cap) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      radii[idx] = 0;
      recs[3 * idx + 0] = z, recs[3 * idx + 1] = z, recs[3 * idx + 2] = z;
    }
    return;
  }
  // this workgroup's SH rows -> LDS (coefficient 0 through my_dc, coefficients >= 1 through my_sh, see sh_to_rgb)
  extern __shared__ float s_sh[];
  const float *my_dc = nullptr, *my_sh = nullptr;
  if (shs != nullptr && colors_precomp == nullptr) {
    const int base = blockIdx.x * blockDim.x, nrows = min((int) blockDim.x, P - base);
    if (shs_rest) {
      const int RL = (M - 1) * 3;
      float* s_dc  = s_sh + blockDim.x * sh_pitch(RL);
      stage_rows_in<PRE_THREADS, NT_SH_LOAD_FWD>(s_sh, shs_rest + (size_t) base * RL, nrows, RL);
      for (int i = threadIdx.x; i < nrows * 3; i += blockDim.x) s_dc[i] = shs[(size_t) base * 3 + i];
      my_dc = s_dc + threadIdx.x * 3, my_sh = s_sh + threadIdx.x * sh_pitch(RL) - 3;
    } else {
      const int RL = M * 3;
      stage_rows_in<PRE_THREADS, NT_SH_LOAD_FWD>(s_sh, shs + (size_t) base * RL, nrows, RL);
      my_dc = my_sh = s_sh + threadIdx.x * sh_pitch(RL);
    }
  }
  // (DK > 0) the deform's tables behind the SH rows: joints and bones.  (Staging the workgroup's logit rows there as well -- a
  // lane gathers K of its row's M logits AFTER the search -- saved nothing at 100k Gaussians and cost the launch a third of
  // its resident waves at 300k and 500k: 11 KB more LDS per workgroup.)
  float *s_j = nullptr, *s_bones = nullptr;
  if constexpr (DK > 0) {
    s_j     = s_sh + dj.lds_offset;
    s_bones = s_j + ((dj.M * 3 + 3) & ~3);
    for (int i = threadIdx.x; i < dj.M * 3; i += PRE_THREADS) s_j[i] = dj.joints[i];
    for (int j = threadIdx.x; j < dj.M; j += PRE_THREADS) load_bone(dj.bone_T, dj.bone_drot, dj.bone_dscale, j, s_bones + j * BONE_F);
  }
  if (threadIdx.x < 16) cam.view[threadIdx.x] = cam_v, cam.proj[threadIdx.x] = cam_p;
  if (threadIdx.x < 3) cam.campos[threadIdx.x] = cam_c;
  __syncthreads();
  if constexpr (DK > 0) {
    if (idx < P) {
      float w[DK], sx[3], sr[4], ss[3];
      int bi[DK];
      const float* row = dj.sp_W + (size_t) idx * dj.M;
      knn_softmax_skin_lane<DK>(dj.M, dj.K, s_j, s_bones, dj_p, [&](int j) { return row[j]; }, w, bi, sx, sr, ss);
      // (straight from the lane: rows through LDS would cost this launch resident workgroups)
#pragma unroll
      for (int k = 0; k < DK; ++k)
        if (k < dj.K) dj.out_weights[(size_t) idx * dj.K + k] = w[k], dj.out_idx[(size_t) idx * dj.K + k] = bi[k];
      deform_activate_lane(dj_p, sx, sr, ss, dj_x, dj_ls, dj_r4, dj_ol, pf_p, pf_s, pf_q, pf_op);
#pragma unroll
      for (int c = 0; c < 3; ++c) dj.means[3 * idx + c] = pf_p[c], dj.scales[3 * idx + c] = pf_s[c];
      reinterpret_cast<float4*>(dj.rotations)[idx] = pf_q;
      dj.opacity[idx]                              = pf_op;
    }
  }
  if constexpr (DK == -1) {
    if (idx < P) {  // deform_forward_kernel<false>: the bones' rows from global memory, in the neighbours' order
      float sx[3] = {0, 0, 0}, sr[4] = {0, 0, 0, 0}, ss[3] = {0, 0, 0};
      const int kmax = dj.largest ? argmax_slot(dj.out_weights + (size_t) idx * dj.K, dj.K) : -1;  // (the row is in cache: just read)
      auto skin = [&](int k, int j, float w) {
        float b[BONE_F];
        load_bone(dj.bone_T, dj.bone_drot, dj.bone_dscale, j, b);
        float y[3];
        se3_act(b, dj_p, y);
        const float wx = kmax < 0 ? w : (k == kmax ? 1.f : 0.f);  // `largest`: the position follows that one bone
        sx[0] += y[0] * wx, sx[1] += y[1] * wx, sx[2] += y[2] * wx;
        sr[0] += b[7] * w, sr[1] += b[8] * w, sr[2] += b[9] * w, sr[3] += b[10] * w;
        ss[0] += b[11] * w, ss[1] += b[12] * w, ss[2] += b[13] * w;
      };
#pragma unroll
      for (int q = 0; q < PREF_K; ++q)
        if (q < dj.K) skin(q, sk_j[q], sk_w[q]);
      for (int k = PREF_K; k < dj.K; ++k) skin(k, (int) dj.out_idx[(size_t) idx * dj.K + k], dj.out_weights[(size_t) idx * dj.K + k]);
      deform_activate_lane(dj_p, sx, sr, ss, dj_x, dj_ls, dj_r4, dj_ol, pf_p, pf_s, pf_q, pf_op);
#pragma unroll
      for (int c = 0; c < 3; ++c) dj.means[3 * idx + c] = pf_p[c], dj.scales[3 * idx + c] = pf_s[c];
      reinterpret_cast<float4*>(dj.rotations)[idx] = pf_q;
      dj.opacity[idx]                              = pf_op;
    }
  }
  if (idx >= P) {
    if (idx < P_cap) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      radii[idx] = 0;
      recs[3 * idx + 0] = z, recs[3 * idx + 1] = z, recs[3 * idx + 2] = z;
    }
    return;
  }

  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;  // culled Gaussians get an all-zero record
  int radius = 0;
  uint32_t clamp_bits = 0;
  int mn[2] = {0, 0}, mx[2] = {0, 0};
  const float p[3] = {pf_p[0], pf_p[1], pf_p[2]};
  float pv[3], ph[4];
  bool ok;
  if (COLMAP) {
    xf3_cm(p, cam.view, pv);
    ok = !(pv[2] <= 0.2f);
  } else {
    xf3_rm(p, cam.view, pv);
    ok = !(pv[2] <= -1.0f);
  }
  if (ok) {
    if (COLMAP)
      xf4_cm(p, cam.proj, ph);
    else
      xf4_rm(p, cam.proj, ph);
    const float p_w = 1.0f / (ph[3] + 0.0000001f);
    const float ppx = ph[0] * p_w, ppy = ph[1] * p_w;
    float c6[6];
    if (cov3D_precomp != nullptr) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = pf_c6[i];
    } else {
      const float s[3] = {pf_s[0], pf_s[1], pf_s[2]};
      const float4 qv  = pf_q;
      const float q[4] = {qv.x, qv.y, qv.z, qv.w};
      if (COLMAP)
        cov3d_cm(s, scale_modifier, q, c6);
      else
        cov3d_rm(s, q, c6);
    }
    float cov[3];
    if (COLMAP) {
      ProjCM pr = proj_cm(p, focal_x, focal_y, tan_fovx, tan_fovy, c6, cam.view);
      M3 c      = m3_mul(m3_mul(m3_t(pr.T), m3_t(pr.V)), pr.T);
      cov[0] = c.m[0][0] + 0.3f, cov[1] = c.m[0][1], cov[2] = c.m[1][1] + 0.3f;
    } else {
      ProjRM pr = proj_rm(p, focal_x, focal_y, tan_fovx, tan_fovy, cam.view);
      cov2d_rm(pr, c6, cov);
    }
    const float det = (cov[0] * cov[2] - cov[1] * cov[1]);
    if (det != 0.0f) {
      const float det_inv = 1.f / det;
      const float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
      const float mid      = 0.5f * (cov[0] + cov[2]);
      const float lambda1  = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda2  = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
      const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      const float pix[2]    = {ndc2pix(ppx, W), ndc2pix(ppy, H)};
      tile_rect(pix[0], pix[1], (int) my_radius, gx, gy, mn, mx);
      const uint32_t area = (uint32_t) (mx[0] - mn[0]) * (uint32_t) (mx[1] - mn[1]);
      if (area != 0) {
        float rgb[3];
        if (colors_precomp == nullptr) {
          sh_to_rgb(D, p, cam.campos, my_dc, my_sh, rgb, &clamp_bits);
        } else {
          rgb[0] = pf_col[0], rgb[1] = pf_col[1], rgb[2] = pf_col[2];
        }
        radius = (int) my_radius;
        r0     = make_float4(pix[0], pix[1], conic[0], conic[1]);
        r1     = make_float4(conic[2], pf_op, rgb[0], rgb[1]);
        // qmax = 2 ln(255 o): a pixel can reach alpha = min(0.99, o * exp(-q/2)) >= 1/255 only where the conic form
        // q(d) <= qmax.  The blend kernels skip a splat for a whole wave when the minimum of q over the wave's pixel
        // rectangle exceeds it (render.hip: splat_reaches_rect); +0.01 absorbs logf / exp rounding.
        const float o255 = 255.0f * pf_op;
        const float qmax = o255 > 1.0f ? 2.0f * logf(o255) + 0.01f : -1.0f;
        r2 = make_float4(rgb[2], pv[2], __int_as_float(radius | (int) (clamp_bits << 28)), qmax);
      }
    }
  }
  radii[idx]        = radius;
  recs[3 * idx + 0] = r0;
  recs[3 * idx + 1] = r1;
  recs[3 * idx + 2] = r2;
  // (per-tile instance counts are accumulated by binning.hip::count_tiles_kernel, 16 lanes per Gaussian)
}

// ------------------------------------------------------------------------------------------------ backward
// dL_dmeans += through dir = normalize(mean - campos)
__device__ __forceinline__ void dnormvdv3(const float* v, const float* dv, float* o) {
  const float sum2     = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

// SH backward: writes dL_dsh[M][3] for this Gaussian and returns dL_dmean contribution.  `sh` and (dL_ddc, dL_dsh)
// follow the convention of sh_to_rgb (coefficient 0 through the first pointer; only coefficients >= 1 of sh are read).
// Every read of `sh` happens before the first write to dL_dsh, so the two may be the SAME row (the kernel keeps the
// coefficients and their gradients in one LDS row per lane).
// `factors` (6 floats, or NULL): instead of the M x 3 gradient row, emit what it is the outer product of -- the unit
// direction (x, y, z) and the clamp-masked colour gradient g -- for skgs_sh_grad_from_factors (view-parallel training).
__device__ __forceinline__ void sh_backward(int deg, int M, const float* mean, const float* campos, const float* sh,
    uint32_t clamp_bits, const float* dL_dcolor, float* dL_ddc, float* dL_dsh, float* dL_dmean_out, float* factors) {
  const float dir_orig[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
  const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  const float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  float g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c] = dL_dcolor[c] * (((clamp_bits >> c) & 1u) ? 0.f : 1.f);
  // ---- reads: d(colour)/d(direction)
  float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#define SHV(i, c) sh[(i) *3 + (c)]
  if (deg > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] = -SH_C1 * SHV(3, c);
      dy[c] = -SH_C1 * SHV(1, c);
      dz[c] = SH_C1 * SHV(2, c);
    }
    if (deg > 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dx[c] += SH_C2[0] * y * SHV(4, c) + SH_C2[2] * 2.f * -x * SHV(6, c) + SH_C2[3] * z * SHV(7, c) + SH_C2[4] * 2.f * x * SHV(8, c);
        dy[c] += SH_C2[0] * x * SHV(4, c) + SH_C2[1] * z * SHV(5, c) + SH_C2[2] * 2.f * -y * SHV(6, c) + SH_C2[4] * 2.f * -y * SHV(8, c);
        dz[c] += SH_C2[1] * y * SHV(5, c) + SH_C2[2] * 2.f * 2.f * z * SHV(6, c) + SH_C2[3] * x * SHV(7, c);
      }
      if (deg > 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dx[c] += (SH_C3[0] * SHV(9, c) * 3.f * 2.f * xy + SH_C3[1] * SHV(10, c) * yz + SH_C3[2] * SHV(11, c) * -2.f * xy +
                    SH_C3[3] * SHV(12, c) * -3.f * 2.f * xz + SH_C3[4] * SHV(13, c) * (-3.f * xx + 4.f * zz - yy) +
                    SH_C3[5] * SHV(14, c) * 2.f * xz + SH_C3[6] * SHV(15, c) * 3.f * (xx - yy));
          dy[c] += (SH_C3[0] * SHV(9, c) * 3.f * (xx - yy) + SH_C3[1] * SHV(10, c) * xz +
                    SH_C3[2] * SHV(11, c) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SHV(12, c) * -3.f * 2.f * yz +
                    SH_C3[4] * SHV(13, c) * -2.f * xy + SH_C3[5] * SHV(14, c) * -2.f * yz + SH_C3[6] * SHV(15, c) * -3.f * 2.f * xy);
          dz[c] += (SH_C3[1] * SHV(10, c) * xy + SH_C3[2] * SHV(11, c) * 4.f * 2.f * yz +
                    SH_C3[3] * SHV(12, c) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SHV(13, c) * 4.f * 2.f * xz +
                    SH_C3[5] * SHV(14, c) * (xx - yy));
        }
      }
    }
  }
#undef SHV
  const float dL_ddir[3] = {dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2], dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2],
      dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2]};
  dnormvdv3(dir_orig, dL_ddir, dL_dmean_out);
  if (factors) {
    factors[0] = x, factors[1] = y, factors[2] = z, factors[3] = g[0], factors[4] = g[1], factors[5] = g[2];
    return;
  }
  // ---- writes: d(colour)/d(coefficient i) = basis_i(direction)   (sh_basis_row below restates these coefficients)
#define SETSH(i, coef)                                             \
  {                                                                \
    const float _k = (coef);                                       \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) ((i) == 0 ? dL_ddc : dL_dsh)[(i) *3 + c] = _k * g[c]; \
  }
  SETSH(0, SH_C0);
  if (deg > 0) {
    SETSH(1, -SH_C1 * y);
    SETSH(2, SH_C1 * z);
    SETSH(3, -SH_C1 * x);
    if (deg > 1) {
      SETSH(4, SH_C2[0] * xy);
      SETSH(5, SH_C2[1] * yz);
      SETSH(6, SH_C2[2] * (2.f * zz - xx - yy));
      SETSH(7, SH_C2[3] * xz);
      SETSH(8, SH_C2[4] * (xx - yy));
      if (deg > 2) {
        SETSH(9, SH_C3[0] * y * (3.f * xx - yy));
        SETSH(10, SH_C3[1] * xy * z);
        SETSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
        SETSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
        SETSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
        SETSH(14, SH_C3[5] * z * (xx - yy));
        SETSH(15, SH_C3[6] * x * (xx - 3.f * yy));
      }
    }
  }
#undef SETSH
  // coefficients above the active degree stay zero
  const int used = (deg + 1) * (deg + 1);
  for (int i = used; i < M; ++i) dL_dsh[i * 3] = 0.f, dL_dsh[i * 3 + 1] = 0.f, dL_dsh[i * 3 + 2] = 0.f;
}

// DBJ: the skeleton stage's deform backward as a job of this launch (skgs_raster_grads.deform_backward_job): the lane that has
// just produced dL/d(mean, scale, rotation, opacity) of its Gaussian hands them, in registers, to deform_bwd_moments
// (deform_lane.h: the body of deform_backward_moments_kernel) -- the Gaussian's parameter gradients, its logit gradients and
// the workgroup's partial bone moments, in the LDS the SH rows have just left.
static_assert(PRE_BWD_THREADS == DEFORM_BWD_THREADS, "the deform backward job runs in this launch's workgroups");
// JOB 0: none; 1 (DBJ): that; 2 / 3: the superpoint stage's ROWS pass (skgs_raster_grads.sp_skinning_job, sp_rows_lane<8 / 0>
// of deform_lane.h = the body of sp_backward_rows_kernel) with its bone table in the same LDS.
template <bool COLMAP, int JOB>
__global__ void __launch_bounds__(PRE_BWD_THREADS) preprocess_backward_kernel(int P, int D, int M, const float* __restrict__ means3D,
    const int32_t* __restrict__ radii, const float* __restrict__ shs, const float* __restrict__ shs_rest,
    const float* __restrict__ scales,
    const float* __restrict__ rotations, float scale_modifier, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos, int W,
    int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, const float4* __restrict__ recs,
    float* __restrict__ gradacc /*[P][16]*/, int moments, int rezero, const float* __restrict__ gin_means2D,
    const float* __restrict__ gin_conic, const float* __restrict__ gin_opacity, int E, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dconic_out, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dsh_rest,
    float* __restrict__ dL_dscales, float* __restrict__ dL_drot, float* __restrict__ dL_dextras,
    float* __restrict__ sh_factors /* [P,6] or NULL: see sh_backward */, const float* __restrict__ tanfov_dev,
    const int32_t* __restrict__ live, float* __restrict__ stat_accum, float* __restrict__ stat_denom,
    float* __restrict__ stat_max_radii, float stat_mult, DeformBwdArgs dbj, SpRowsArgs srj) {
  constexpr bool DBJ = JOB == 1;
  if (live) P = min(P, live[0]);  // the number of Gaussians is a device word: one captured graph survives densification
  if ((int) (blockIdx.x * blockDim.x) >= P) {  // a workgroup of the capacity's slack rows (before any barrier)
    if constexpr (DBJ) deform_bwd_zero_partials(dbj);
    return;
  }
  if (tanfov_dev) {
    tan_fovx = tanfov_dev[0], tan_fovy = tanfov_dev[1];
    focal_x = W / (2.0f * tan_fovx), focal_y = H / (2.0f * tan_fovy);
  }
  __shared__ Cam cam;
  if (threadIdx.x < 16) {
    cam.view[threadIdx.x] = viewmatrix[threadIdx.x];
    cam.proj[threadIdx.x] = projmatrix[threadIdx.x];
  }
  if (threadIdx.x < 3) cam.campos[threadIdx.x] = campos[threadIdx.x];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // SH rows of this workgroup -> LDS; the gradient rows are built in the same LDS rows (sh_backward reads before it
  // writes) and leave as one contiguous span at the end.  The DC term is not read by the backward.
  extern __shared__ float s_sh[];
  const bool staged = shs != nullptr && (dL_dsh != nullptr || sh_factors != nullptr);
  const int RL      = shs_rest ? (M - 1) * 3 : M * 3;
  const int base = blockIdx.x * blockDim.x, nrows = min((int) blockDim.x, P - base);
  // Every per-Gaussian input is requested BEFORE the SH staging barrier, unconditionally: radius -> gradient row ->
  // record -> mean / scale / rotation used to be a chain of dependent round trips behind `visible` (the kernel runs at
  // ~1.5 waves per SIMD: its duration is the length of one lane's dependency chain).
  int pf_radius = 0;
  float4 pf_row[4], pf_rec[3], pf_q = make_float4(0.f, 0.f, 0.f, 1.f);
  float pf_p[3] = {0.f, 0.f, 0.f}, pf_s[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) pf_row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  pf_rec[0] = pf_rec[1] = pf_rec[2] = pf_row[0];
  if (idx < P) {
    pf_radius = radii[idx];
    const float* row = gradacc + (size_t) idx * GRAD_ROW;
#pragma unroll
    for (int i = 0; i < 4; ++i) pf_row[i] = stream_load4<NT_GRADROW_LOAD>(row + 4 * i);
    pf_rec[0] = recs[3 * idx], pf_rec[1] = recs[3 * idx + 1], pf_rec[2] = recs[3 * idx + 2];
    pf_p[0] = means3D[3 * idx], pf_p[1] = means3D[3 * idx + 1], pf_p[2] = means3D[3 * idx + 2];
    if (scales) {
      pf_s[0] = scales[3 * idx], pf_s[1] = scales[3 * idx + 1], pf_s[2] = scales[3 * idx + 2];
      pf_q = reinterpret_cast<const float4*>(rotations)[idx];
    }
  }
  DeformBwdLane dbl;  // (DBJ) the deform backward's own per-Gaussian inputs ride in the same round trip
  if constexpr (DBJ) deform_bwd_prefetch(dbj, idx, idx < P, dbl);
  float dj_gm[3] = {0.f, 0.f, 0.f}, dj_gs[3] = {0.f, 0.f, 0.f}, dj_go = 0.f;
  float4 dj_gr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (staged) stage_rows_in<PRE_BWD_THREADS, NT_SH_LOAD_BWD>(s_sh, (shs_rest ? shs_rest : shs) + (size_t) base * RL, nrows, RL);
  __syncthreads();
  if (idx < P) {

  // gradients accumulated by the blend backward (+ the optional chained-in ones)
  float gm2[2] = {0.f, 0.f}, gcon[3] = {0.f, 0.f, 0.f}, gop = 0.f, gcol[3] = {0.f, 0.f, 0.f}, gex[4] = {0, 0, 0, 0};
  const bool visible = pf_radius > 0;
  {
    float4* row = reinterpret_cast<float4*>(gradacc + (size_t) idx * GRAD_ROW);
    float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
    if (visible) {
      a = pf_row[0], b = pf_row[1], c = pf_row[2], d = pf_row[3];
      // skgs_raster_grads::workspace_is_zero: hand the scratch back all zero (rows of culled Gaussians are never touched)
      if (rezero) row[0] = row[1] = row[2] = row[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    gm2[0] = a.x, gm2[1] = a.y, gcon[0] = a.z, gcon[1] = a.w, gcon[2] = b.x, gop = b.y, gcol[0] = b.z, gcol[1] = b.w;
    gcol[2] = c.x, gex[0] = c.y, gex[1] = c.z, gex[2] = c.w, gex[3] = d.x;
    if (moments) {
      // fast blend build: slots 0..4 are sum gA {dx, dy, dx^2, dx dy, dy^2} over all pixels (render_blend.inl);
      // dL/dmean2D = -(conic . [m1, m2]) * 0.5 * (W, H), dL/dconic = -0.5 * [m3, m4, m5]
      // (the rows hold the moments of gA = G dL/dalpha; the opacity factor of w = o gA is applied here, once)
      const float4 q0 = pf_rec[0], q1 = pf_rec[1];
      const float cx = q0.z, cy = q0.w, cz = q1.x, o = q1.y;
      const float m1 = o * a.x, m2 = o * a.y;
      gm2[0]  = -(cx * m1 + cy * m2) * (0.5f * W);
      gm2[1]  = -(cz * m2 + cy * m1) * (0.5f * H);
      gcon[0] = -0.5f * (o * a.z), gcon[1] = -0.5f * (o * a.w), gcon[2] = -0.5f * (o * b.x);
    }
  }
  if (gin_means2D) gm2[0] += gin_means2D[3 * idx], gm2[1] += gin_means2D[3 * idx + 1];
  if (gin_conic) gcon[0] += gin_conic[4 * idx], gcon[1] += gin_conic[4 * idx + 1], gcon[2] += gin_conic[4 * idx + 3];
  if (gin_opacity) gop += gin_opacity[idx];
  dL_dmeans2D[3 * idx] = gm2[0], dL_dmeans2D[3 * idx + 1] = gm2[1];
  dL_dmeans2D[3 * idx + 2] = gin_means2D ? gin_means2D[3 * idx + 2] : 0.f;
  // densification statistics of this view (skgs_densify_stats folded in: the arithmetic of densify.hip::densify_stats_kernel
  // on the values just written -- one launch less per training step)
  if (stat_accum && pf_radius > 0) {
    stat_max_radii[idx] = fmaxf(stat_max_radii[idx], (float) pf_radius);
    const float nrm     = sqrtf(gm2[0] * gm2[0] + gm2[1] * gm2[1]);
    stat_accum[idx]     = stat_accum[idx] + (stat_mult == 1.0f ? nrm : stat_mult * nrm);
    stat_denom[idx]     = stat_denom[idx] + 1.0f;
  }
  if (dL_dconic_out) {
    dL_dconic_out[4 * idx] = gcon[0], dL_dconic_out[4 * idx + 1] = gcon[1];
    dL_dconic_out[4 * idx + 2] = gin_conic ? gin_conic[4 * idx + 2] : 0.f;
    dL_dconic_out[4 * idx + 3] = gcon[2];
  }
  // (NULL with a job attached: the job takes these per-Gaussian gradients over in registers, nobody reads the arrays)
  if (dL_dopacity) dL_dopacity[idx] = gop;
  if (dL_dcolors) dL_dcolors[3 * idx] = gcol[0], dL_dcolors[3 * idx + 1] = gcol[1], dL_dcolors[3 * idx + 2] = gcol[2];
  for (int e = 0; e < E; ++e) dL_dextras[(size_t) idx * E + e] = gex[e];

  float gmean[3] = {0.f, 0.f, 0.f}, gcov[6] = {0, 0, 0, 0, 0, 0}, gscale[3] = {0, 0, 0}, grot[4] = {0, 0, 0, 0};
  // gradient rows of the SH coefficients: one [M][3] row, or split DC [1][3] / rest [M-1][3] rows (rest biased by -3)
  float* my_row  = s_sh + threadIdx.x * sh_pitch(RL);
  float* gsh_dc  = !dL_dsh ? nullptr : (dL_dsh_rest ? dL_dsh + (size_t) idx * 3 : my_row);
  float* gsh_row = shs_rest ? my_row - 3 : my_row;  // the staged coefficients; the gradient row replaces them in place
  float* fac     = sh_factors ? sh_factors + (size_t) idx * 6 : nullptr;
  if (!visible) {
    if (fac) {
#pragma unroll
      for (int i = 0; i < 6; ++i) fac[i] = 0.f;
    } else if (gsh_dc) {
      for (int i = 0; i < 3; ++i) gsh_dc[i] = 0.f;
      for (int i = 3; i < M * 3; ++i) gsh_row[i] = 0.f;
    }
  } else {
    const float p[3] = {pf_p[0], pf_p[1], pf_p[2]};
    float c6[6];
    float s[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};
    if (scales) {
      s[0] = pf_s[0], s[1] = pf_s[1], s[2] = pf_s[2];
      q[0] = pf_q.x, q[1] = pf_q.y, q[2] = pf_q.z, q[3] = pf_q.w;
    }
    if (cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = cov3D_precomp[6 * idx + i];
    } else if (COLMAP) {
      cov3d_cm(s, scale_modifier, q, c6);  // recomputed (same arithmetic as the forward) instead of stored
    } else {
      cov3d_rm(s, q, c6);
    }
    const float h_x = focal_x, h_y = focal_y;
    // ---- conic -> cov2D -> cov3D, mean (part 1: ASSIGNED) ----
    if (COLMAP) {
      ProjCM pr = proj_cm(p, h_x, h_y, tan_fovx, tan_fovy, c6, cam.view);
      M3 c2     = m3_mul(m3_mul(m3_t(pr.T), m3_t(pr.V)), pr.T);
      const float a = c2.m[0][0] + 0.3f, b = c2.m[0][1], c = c2.m[1][1] + 0.3f;
      const float denom = a * c - b * b;
      float dL_da = 0, dL_db = 0, dL_dc = 0;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) pr.T.m[i][j]
#define VV(i, j) pr.V.m[i][j]
#define WW(i, j) pr.W.m[i][j]
      if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * gcon[0] + 2 * b * c * gcon[1] + (denom - a * c) * gcon[2]);
        dL_dc = denom2inv * (-a * a * gcon[2] + 2 * a * b * gcon[1] + (denom - a * c) * gcon[0]);
        dL_db = denom2inv * 2 * (b * c * gcon[0] - (denom + 2 * b * b) * gcon[1] + a * b * gcon[2]);
        gcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
        gcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
        gcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
        gcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
        gcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
        gcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
      }
      const float dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                         (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
      const float dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                         (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
      const float dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                         (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
      const float dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                         (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
      const float dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                         (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
      const float dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                         (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
      const float dJ00 = WW(0, 0) * dT00 + WW(0, 1) * dT01 + WW(0, 2) * dT02;
      const float dJ02 = WW(2, 0) * dT00 + WW(2, 1) * dT01 + WW(2, 2) * dT02;
      const float dJ11 = WW(1, 0) * dT10 + WW(1, 1) * dT11 + WW(1, 2) * dT12;
      const float dJ12 = WW(2, 0) * dT10 + WW(2, 1) * dT11 + WW(2, 2) * dT12;
#undef TT
#undef VV
#undef WW
      const float tz = 1.f / pr.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      const float dt[3] = {pr.xm * -h_x * tz2 * dJ02, pr.ym * -h_y * tz2 * dJ12,
          -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * pr.t[0]) * tz3 * dJ02 + (2 * h_y * pr.t[1]) * tz3 * dJ12};
      const float* vm = cam.view;
      gmean[0] = vm[0] * dt[0] + vm[1] * dt[1] + vm[2] * dt[2];
      gmean[1] = vm[4] * dt[0] + vm[5] * dt[1] + vm[6] * dt[2];
      gmean[2] = vm[8] * dt[0] + vm[9] * dt[1] + vm[10] * dt[2];
    } else {
      // literal restatement of the row-major variant INCLUDING its self-inconsistencies
      // (gaussian_preprocess.cu:241 uses T[1]; dL_dT2/5/8 = 0 so dL_dtx = dL_dty = 0)
      ProjRM pr = proj_rm(p, h_x, h_y, tan_fovx, tan_fovy, cam.view);
      float cov[3];
      cov2d_rm(pr, c6, cov);
      const float* T = pr.T;
      const float* Wm = pr.W;
      const float a = cov[0], b = cov[1], c = cov[2];
      const float denom = a * c - b * b;
      float dL_da = 0, dL_db = 0, dL_dc = 0;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
      if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * gcon[0] + 2 * b * c * gcon[1] + (denom - a * c) * gcon[2]);
        dL_dc = denom2inv * (-a * a * gcon[2] + 2 * a * b * gcon[1] + (denom - a * c) * gcon[0]);
        dL_db = denom2inv * 2 * (b * c * gcon[0] - (denom + 2 * b * b) * gcon[1] + a * b * gcon[2]);
        gcov[0] = (T[0] * T[0] * dL_da + T[0] * T[1] * dL_db + T[1] * T[1] * dL_dc);
        gcov[3] = (T[1] * T[1] * dL_da + T[1] * T[4] * dL_db + T[4] * T[4] * dL_dc);
        gcov[5] = (T[6] * T[6] * dL_da + T[6] * T[7] * dL_db + T[7] * T[7] * dL_dc);
        gcov[1] = 2 * T[0] * T[3] * dL_da + (T[0] * T[4] + T[3] * T[1]) * dL_db + 2 * T[1] * T[4] * dL_dc;
        gcov[2] = 2 * T[0] * T[6] * dL_da + (T[0] * T[7] + T[6] * T[1]) * dL_db + 2 * T[1] * T[7] * dL_dc;
        gcov[4] = 2 * T[6] * T[3] * dL_da + (T[3] * T[7] + T[6] * T[4]) * dL_db + 2 * T[4] * T[7] * dL_dc;
      }
      const float dT0 = 2 * (T[0] * c6[0] + T[3] * c6[1] + T[6] * c6[2]) * dL_da + (T[1] * c6[0] + T[4] * c6[1] + T[7] * c6[2]) * dL_db;
      const float dT1 = 2 * (T[1] * c6[0] + T[4] * c6[1] + T[7] * c6[2]) * dL_dc + (T[0] * c6[0] + T[3] * c6[1] + T[6] * c6[2]) * dL_db;
      const float dT2 = 0;
      const float dT3 = 2 * (T[0] * c6[1] + T[3] * c6[3] + T[6] * c6[4]) * dL_da + (T[1] * c6[1] + T[4] * c6[3] + T[7] * c6[4]) * dL_db;
      const float dT4 = 2 * (T[1] * c6[1] + T[4] * c6[3] + T[7] * c6[4]) * dL_dc + (T[0] * c6[3] + T[3] * c6[4] + T[6] * c6[5]) * dL_db;
      const float dT5 = 0;
      const float dT6 = 2 * (T[0] * c6[2] + T[3] * c6[4] + T[6] * c6[5]) * dL_da + (T[1] * c6[2] + T[4] * c6[4] + T[7] * c6[5]) * dL_db;
      const float dT7 = 2 * (T[1] * c6[2] + T[4] * c6[4] + T[7] * c6[5]) * dL_dc + (T[0] * c6[2] + T[3] * c6[4] + T[6] * c6[5]) * dL_db;
      const float dT8 = 0;
      const float dJ00 = Wm[0] * dT0 + Wm[3] * dT3 + Wm[6] * dT6;
      const float dJ02 = Wm[0] * dT2 + Wm[3] * dT5 + Wm[6] * dT8;
      const float dJ11 = Wm[1] * dT1 + Wm[4] * dT4 + Wm[7] * dT7;
      const float dJ12 = Wm[1] * dT2 + Wm[4] * dT5 + Wm[7] * dT8;
      const float tz = 1.f / pr.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      const float dt[3] = {pr.xm * -h_x * tz2 * dJ02, pr.ym * -h_y * tz2 * dJ12,
          -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * pr.t[0]) * tz3 * dJ02 + (2 * h_y * pr.t[1]) * tz3 * dJ12};
      const float* vm = cam.view;
      gmean[0] = vm[0] * dt[0] + vm[4] * dt[1] + vm[8] * dt[2];
      gmean[1] = vm[1] * dt[0] + vm[5] * dt[1] + vm[9] * dt[2];
      gmean[2] = vm[2] * dt[0] + vm[6] * dt[1] + vm[10] * dt[2];
    }
    // ---- projection of the 2D mean (part 2: +=) ----
    {
      const float* proj = cam.proj;
      float mh[4], d2[3];
      if (COLMAP) {
        xf4_cm(p, proj, mh);
        const float m_w  = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * m_w * m_w;
        d2[0] = (proj[0] * m_w - proj[3] * mul1) * gm2[0] + (proj[1] * m_w - proj[3] * mul2) * gm2[1];
        d2[1] = (proj[4] * m_w - proj[7] * mul1) * gm2[0] + (proj[5] * m_w - proj[7] * mul2) * gm2[1];
        d2[2] = (proj[8] * m_w - proj[11] * mul1) * gm2[0] + (proj[9] * m_w - proj[11] * mul2) * gm2[1];
      } else {
        xf4_rm(p, proj, mh);
        const float m_w  = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = (proj[0] * p[0] + proj[1] * p[1] + proj[2] * p[2] + proj[3]) * m_w * m_w;
        const float mul2 = (proj[4] * p[0] + proj[5] * p[1] + proj[6] * p[2] + proj[7]) * m_w * m_w;
        d2[0] = (proj[0] * m_w - proj[12] * mul1) * gm2[0] + (proj[4] * m_w - proj[12] * mul2) * gm2[1];
        d2[1] = (proj[1] * m_w - proj[13] * mul1) * gm2[0] + (proj[5] * m_w - proj[13] * mul2) * gm2[1];
        d2[2] = (proj[2] * m_w - proj[14] * mul1) * gm2[0] + (proj[6] * m_w - proj[14] * mul2) * gm2[1];
      }
      gmean[0] += d2[0], gmean[1] += d2[1], gmean[2] += d2[2];
    }
    // ---- SH (part 3: +=) ----
    if (shs) {
      const uint32_t clamp_bits = (__float_as_uint(pf_rec[2].z) >> 28) & 7u;
      float dm[3];
      sh_backward(D, M, p, cam.campos, gsh_row, clamp_bits, gcol, gsh_dc, gsh_row, dm, fac);  // coefficients and gradients share the row
      gmean[0] += dm[0], gmean[1] += dm[1], gmean[2] += dm[2];
    }
    // ---- Sigma3D -> scale, rotation ----
    if (scales) {
      if (COLMAP) {
        const float x = q[0], y = q[1], z = q[2], r = q[3];
        M3 R           = rot_cm(q);
        const float sm[3] = {scale_modifier * s[0], scale_modifier * s[1], scale_modifier * s[2]};
        M3 Mm          = scale_rot_cm(sm, R);
        M3 dSg;
        dSg.m[0][0] = gcov[0], dSg.m[0][1] = 0.5f * gcov[1], dSg.m[0][2] = 0.5f * gcov[2];
        dSg.m[1][0] = 0.5f * gcov[1], dSg.m[1][1] = gcov[3], dSg.m[1][2] = 0.5f * gcov[4];
        dSg.m[2][0] = 0.5f * gcov[2], dSg.m[2][1] = 0.5f * gcov[4], dSg.m[2][2] = gcov[5];
        M3 M2;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int w = 0; w < 3; ++w) M2.m[c][w] = Mm.m[c][w] * 2.0f;
        M3 dM  = m3_mul(M2, dSg);
        M3 Rt  = m3_t(R);
        M3 dMt = m3_t(dM);
#pragma unroll
        for (int k = 0; k < 3; ++k) gscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int w = 0; w < 3; ++w) dMt.m[k][w] *= sm[k];
#define DD(i, j) dMt.m[i][j]
        grot[0] = 2 * y * (DD(1, 0) + DD(0, 1)) + 2 * z * (DD(2, 0) + DD(0, 2)) + 2 * r * (DD(1, 2) - DD(2, 1)) - 4 * x * (DD(2, 2) + DD(1, 1));
        grot[1] = 2 * x * (DD(1, 0) + DD(0, 1)) + 2 * r * (DD(2, 0) - DD(0, 2)) + 2 * z * (DD(1, 2) + DD(2, 1)) - 4 * y * (DD(2, 2) + DD(0, 0));
        grot[2] = 2 * r * (DD(0, 1) - DD(1, 0)) + 2 * x * (DD(2, 0) + DD(0, 2)) + 2 * y * (DD(1, 2) + DD(2, 1)) - 4 * z * (DD(1, 1) + DD(0, 0));
        grot[3] = 2 * z * (DD(0, 1) - DD(1, 0)) + 2 * y * (DD(2, 0) - DD(0, 2)) + 2 * x * (DD(1, 2) - DD(2, 1));
#undef DD
      } else {
        float R[9];
        q2R_rm(q, R);
        const float* g = gcov;
        gscale[0] = R[0] * R[0] * g[0] + R[0] * R[3] * g[1] + R[0] * R[6] * g[2] + R[3] * R[3] * g[3] + R[3] * R[6] * g[4] + R[6] * R[6] * g[5];
        gscale[1] = R[1] * R[1] * g[0] + R[1] * R[4] * g[1] + R[1] * R[7] * g[2] + R[4] * R[4] * g[3] + R[4] * R[7] * g[4] + R[7] * R[7] * g[5];
        gscale[2] = R[2] * R[2] * g[0] + R[2] * R[5] * g[1] + R[2] * R[8] * g[2] + R[5] * R[5] * g[3] + R[5] * R[8] * g[4] + R[8] * R[8] * g[5];
        gscale[0] *= 2 * s[0], gscale[1] *= 2 * s[1], gscale[2] *= 2 * s[2];
        const float sx2 = s[0] * s[0], sy2 = s[1] * s[1], sz2 = s[2] * s[2];
        float dR[9];
        dR[0] = (2 * R[0] * g[0] + R[3] * g[1] + R[6] * g[2]) * sx2;
        dR[1] = (2 * R[1] * g[0] + R[4] * g[1] + R[7] * g[2]) * sy2;
        dR[2] = (2 * R[2] * g[0] + R[5] * g[1] + R[8] * g[2]) * sz2;
        dR[3] = (2 * R[3] * g[3] + R[0] * g[1] + R[6] * g[4]) * sx2;
        dR[4] = (2 * R[4] * g[3] + R[1] * g[1] + R[7] * g[4]) * sy2;
        dR[5] = (2 * R[5] * g[3] + R[2] * g[1] + R[8] * g[4]) * sz2;
        dR[6] = (2 * R[6] * g[5] + R[0] * g[2] + R[3] * g[4]) * sx2;
        dR[7] = (2 * R[7] * g[5] + R[1] * g[2] + R[4] * g[4]) * sy2;
        dR[8] = (2 * R[8] * g[5] + R[2] * g[2] + R[5] * g[4]) * sz2;
        const float x = q[0], y = q[1], z = q[2], w = q[3];
        grot[0] = 2 * (-2 * x * (dR[4] + dR[8]) + y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) + w * (dR[7] - dR[5]));
        grot[1] = 2 * (x * (dR[1] + dR[3]) - 2 * y * (dR[0] + dR[8]) + z * (dR[5] + dR[7]) + w * (dR[2] - dR[6]));
        grot[2] = 2 * (x * (dR[2] + dR[6]) + y * (dR[5] + dR[7]) - 2 * z * (dR[0] + dR[4]) + w * (dR[3] - dR[1]));
        grot[3] = 2 * (x * (dR[7] - dR[5]) + y * (dR[2] - dR[6]) + z * (dR[3] - dR[1]));
      }
    }
  }
  if (dL_dmeans3D) dL_dmeans3D[3 * idx] = gmean[0], dL_dmeans3D[3 * idx + 1] = gmean[1], dL_dmeans3D[3 * idx + 2] = gmean[2];
  if (dL_dcov3D) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dL_dcov3D[6 * idx + i] = gcov[i];
  }
  if (dL_dscales) dL_dscales[3 * idx] = gscale[0], dL_dscales[3 * idx + 1] = gscale[1], dL_dscales[3 * idx + 2] = gscale[2];
  if (dL_drot) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(grot[0], grot[1], grot[2], grot[3]);
  if constexpr (JOB != 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dj_gm[c] = gmean[c], dj_gs[c] = gscale[c];
    dj_gr = make_float4(grot[0], grot[1], grot[2], grot[3]), dj_go = gop;
  }
  }  // idx < P
  if (staged && dL_dsh) {
    __syncthreads();
    stage_rows_out<PRE_BWD_THREADS, NT_PRE_BWD>((dL_dsh_rest ? dL_dsh_rest : dL_dsh) + (size_t) base * RL, s_sh, nrows, RL);
  }
  if constexpr (DBJ) {
    __syncthreads();  // the SH rows have left the LDS: it is the deform backward's now
    deform_bwd_moments(dbj, P, s_sh, dbl, dj_gm, dj_gs, dj_gr, dj_go);
  }
  if constexpr (JOB >= 2) {
    __syncthreads();
    for (int j = threadIdx.x; j < srj.M; j += PRE_BWD_THREADS) load_bone(srj.bone_T, srj.bone_drot, srj.bone_dscale, j, s_sh + j * BONE_F);
    __syncthreads();
    if (idx < P) sp_rows_lane<(JOB == 2 ? 8 : 0)>(srj, s_sh, idx, dj_gm, dj_gs, dj_gr, dj_go);
  }
}

// dL/dsh of `n_views` views from their factors (sh_backward): row i of Gaussian p = sum_v basis_i(dir_v) * g_v, views
// added in index order (every rank of a view-parallel job holds the same [n_views][P][6] array after the all-gather and
// gets the same bits).  The coefficients are those of sh_backward's SETSH lines, evaluated the same way, so one view
// reproduces the dense row exactly.  Rows leave through LDS as one contiguous span per workgroup.
__device__ __forceinline__ void sh_basis_row(int deg, float x, float y, float z, float* k /*[16]*/) {
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i] = 0.f;
  k[0] = SH_C0;
  if (deg > 0) {
    k[1] = -SH_C1 * y, k[2] = SH_C1 * z, k[3] = -SH_C1 * x;
    if (deg > 1) {
      k[4] = SH_C2[0] * xy, k[5] = SH_C2[1] * yz, k[6] = SH_C2[2] * (2.f * zz - xx - yy), k[7] = SH_C2[3] * xz;
      k[8] = SH_C2[4] * (xx - yy);
      if (deg > 2) {
        k[9]  = SH_C3[0] * y * (3.f * xx - yy);
        k[10] = SH_C3[1] * xy * z;
        k[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        k[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
        k[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        k[14] = SH_C3[5] * z * (xx - yy);
        k[15] = SH_C3[6] * x * (xx - 3.f * yy);
      }
    }
  }
}

__global__ void __launch_bounds__(PRE_THREADS) sh_grad_from_factors_kernel(int P, int n_views, int D, int M,
    const float* __restrict__ factors /*[n_views][P][6]*/, float* __restrict__ dL_dsh /*[P,M,3] or the DC part [P,1,3]*/,
    float* __restrict__ dL_dsh_rest /*[P,M-1,3] or NULL*/) {
  extern __shared__ float s_rows[];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int RL  = dL_dsh_rest ? (M - 1) * 3 : M * 3;
  const int base = blockIdx.x * blockDim.x, nrows = min((int) blockDim.x, P - base);
  float* my_row = s_rows + threadIdx.x * sh_pitch(RL);
  if (idx < P) {
    float acc[16][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
    for (int v = 0; v < n_views; ++v) {
      const float* f = factors + ((size_t) v * P + idx) * 6;
      const float g[3] = {f[3], f[4], f[5]};
      float k[16];
      sh_basis_row(D, f[0], f[1], f[2], k);
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[i][c] = n_views == 1 ? k[i] * g[c] : acc[i][c] + k[i] * g[c];
    }
    const int used = (D + 1) * (D + 1);
    float* dc  = dL_dsh_rest ? dL_dsh + (size_t) idx * 3 : my_row;
    float* row = dL_dsh_rest ? my_row - 3 : my_row;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < M) {
        float* dst = i == 0 ? dc : row + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c] = i < used ? acc[i][c] : 0.f;
      }
    }
    for (int i = 16; i < M; ++i) row[i * 3] = row[i * 3 + 1] = row[i * 3 + 2] = 0.f;
  }
  __syncthreads();
  stage_rows_out((dL_dsh_rest ? dL_dsh_rest : dL_dsh) + (size_t) base * RL, s_rows, nrows, RL);
}

__global__ void mark_visible_kernel(int P, const float* means, const float* view, int colmap, uint8_t* present) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
  const float p[3] = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  float pv[3];
  if (colmap) {
    xf3_cm(p, view, pv);
    present[idx] = !(pv[2] <= 0.2f);
  } else {
    xf3_rm(p, view, pv);
    present[idx] = !(pv[2] <= -1.0f);
  }
}

}  // namespace

int launch_preprocess_forward(const skgs_raster_inputs& in, GeomView g, ImgView im, int32_t* radii, hipStream_t s) {
  const int P = in.P;
  const float focal_y = in.image_height / (2.0f * in.tanfovy);
  const float focal_x = in.image_width / (2.0f * in.tanfovx);
  if (P == 0) return fill_u32(im.tile_counts, 0u, (size_t) im.T, s) || fill_u32(im.cursors, 0u, (size_t) im.T, s);
  ProfScope prof(K_PREPROCESS_FWD, s);
  const bool bucket = in.tile_bucket_capacity > 0;
  dim3 grid((P + PRE_THREADS - 1) / PRE_THREADS), block(PRE_THREADS);
  size_t lds = 0;
  if (in.sh && !in.colors_precomp) {
    const int M = in.sh_coeffs;
    lds = in.sh_rest ? ((size_t) PRE_THREADS * (((M - 1) * 3) | 1) + (size_t) PRE_THREADS * 3) * 4
                     : (size_t) PRE_THREADS * ((M * 3) | 1) * 4;
  }
  KnnDeformJob dj{};
  int dk = 0;
  if (const skgs_knn_deform_job* j = in.deform_job) {
    // the deform in front of the pass: its outputs ARE this pass's per-Gaussian inputs
    const bool search = j->joints != nullptr;  // (joints == NULL: weights / indices are inputs, the skinning alone)
    if (!(j->points && (!search || j->sp_W) && j->bone_T && j->bone_drot && j->bone_dscale && j->xyz && j->log_scale && j->rot &&
            j->opacity_logit && j->out_idx && j->out_weights && j->means && j->scales && j->rotations && j->opacity))
      return set_error("deform_job: NULL pointer");
    if (j->means != in.means3D || j->scales != in.scales || j->rotations != in.rotations || j->opacity != in.opacity)
      return set_error("deform_job: means / scales / rotations / opacity must be the rasterizer's means3D / scales / rotations / opacity");
    if (in.cov3D_precomp) return set_error("deform_job: not with cov3D_precomp");
    if (search && (j->K < 1 || j->K > 8 || j->K > j->M || j->M > SKGS_FUSED_LBS_MAX_BONES))
      return set_error("deform_job: needs 1 <= K <= min(8, M), M <= %d (got K = %d, M = %d)", SKGS_FUSED_LBS_MAX_BONES, j->K, j->M);
    if (!search && (j->K < 1 || j->K > 16 || j->M < 1 || in.live_count))
      return set_error("deform_job (skinning alone): needs 1 <= K <= 16, M >= 1, no row capacity (got K = %d, M = %d)", j->K, j->M);
    if (search && j->largest) return set_error("deform_job: `largest` applies to the skinning alone (joints == NULL)");
    dk = !search ? -1 : j->K <= 5 ? 5 : 8;
    dj.M = j->M, dj.K = j->K, dj.largest = j->largest ? 1 : 0;
    dj.lds_offset = (int) ((lds / 4 + 3) & ~(size_t) 3);
    dj.points = j->points, dj.joints = j->joints, dj.sp_W = j->sp_W, dj.bone_T = j->bone_T, dj.bone_drot = j->bone_drot;
    dj.bone_dscale = j->bone_dscale, dj.xyz = j->xyz, dj.log_scale = j->log_scale, dj.rot = j->rot;
    dj.opacity_logit = j->opacity_logit, dj.out_idx = j->out_idx, dj.out_weights = j->out_weights, dj.means = j->means;
    dj.scales = j->scales, dj.rotations = j->rotations, dj.opacity = j->opacity;
    // (26 KB per workgroup with degree-3 SH rows: six workgroups per CU, as without the job.  The first version kept 44 KB --
    // three per CU = 768 resident workgroups for a grid of 782 at 100k Gaussians: the last 14 ran alone behind the rest,
    // 32 us instead of 22)
    if (search) lds = ((size_t) dj.lds_offset + (size_t) ((j->M * 3 + 3) & ~3) + ((j->M * BONE_F + 3) & ~3)) * 4;
  }
#define SKGS_PRE_FWD(COLMAP_, DK_)                                                                                          \
  hipLaunchKernelGGL((preprocess_forward_kernel<COLMAP_, DK_>), grid, block, lds, s, P, in.sh_degree, in.sh_coeffs, in.means3D, \
      in.scales, in.scale_modifier, in.rotations, in.opacity, in.sh, in.sh_rest, in.cov3D_precomp, in.colors_precomp,        \
      in.viewmatrix, in.projmatrix, in.campos, in.image_width, in.image_height, in.tanfovx, in.tanfovy, focal_x, focal_y,    \
      im.tiles_x, im.tiles_y, radii, g.recs, bucket ? im.cursors : im.tile_counts, bucket ? g.hdr : nullptr,                 \
      in.tanfov_device, in.live_count, dj)
  if (in.colmap) {
    if (dk == 0)
      SKGS_PRE_FWD(true, 0);
    else if (dk == -1)
      SKGS_PRE_FWD(true, -1);
    else if (dk == 5)
      SKGS_PRE_FWD(true, 5);
    else
      SKGS_PRE_FWD(true, 8);
  } else {
    if (dk == 0)
      SKGS_PRE_FWD(false, 0);
    else if (dk == -1)
      SKGS_PRE_FWD(false, -1);
    else if (dk == 5)
      SKGS_PRE_FWD(false, 5);
    else
      SKGS_PRE_FWD(false, 8);
  }
#undef SKGS_PRE_FWD
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_preprocess_backward(const skgs_raster_inputs& in, GeomView g, const int32_t* radii, const skgs_raster_grads& gr,
    hipStream_t s) {
  const int P = in.P;
  if (P == 0) return 0;
  const float focal_y = in.image_height / (2.0f * in.tanfovy);
  const float focal_x = in.image_width / (2.0f * in.tanfovx);
  ProfScope prof(K_PREPROCESS_BWD, s);
  dim3 grid((P + PRE_BWD_THREADS - 1) / PRE_BWD_THREADS), block(PRE_BWD_THREADS);
  size_t lds = (in.sh && (gr.dL_dsh || gr.dL_dsh_factors)) ? (size_t) PRE_BWD_THREADS * (((in.sh_rest ? in.sh_coeffs - 1 : in.sh_coeffs) * 3) | 1) * 4 : 0;
  const int E = (in.extras && gr.dL_dout_extra && gr.dL_dextras) ? in.E : 0;
  DeformBwdArgs dbj{};
  if (const skgs_deform_backward_job* j = gr.deform_backward_job) {  // (arguments checked by skgs_rasterize_backward)
    const skgs_deform_inputs& d = *j->in;
    dbj = DeformBwdArgs{d.K, d.M, d.points, d.weights, d.indices, d.bone_T, d.bone_drot, d.bone_dscale, d.log_scale, d.rot,
        d.opacity_logit, nullptr, j->g_xyz, j->g_log_scale, j->g_rot, j->g_opacity_logit, reinterpret_cast<float*>(j->workspace),
        j->g_sp_W, j->g_logits};
    lds = std::max(lds, deform_bwd_lds_bytes(d.M));
  }
  SpRowsArgs srj{};
  if (const skgs_sp_skinning_job* j = gr.sp_skinning_job) {  // (checked by skgs_rasterize_backward)
    srj = sp_rows_args(*j);
    lds = std::max(lds, sp_rows_lds_bytes(j->in->M));
  }
#define SKGS_PB_ARGS                                                                                                     \
  P, in.sh_degree, in.sh_coeffs, in.means3D, radii, in.sh, in.sh_rest, in.scales, in.rotations, in.scale_modifier,        \
      in.cov3D_precomp, in.viewmatrix, in.projmatrix, in.campos, in.image_width, in.image_height, in.tanfovx, in.tanfovy, \
      focal_x, focal_y, g.recs, gr.workspace, (int) gradacc_rows_hold_moments(), (int) (gr.workspace_is_zero != 0),       \
      gr.grad_means2D_in, gr.grad_conic_in, gr.grad_opacity_in, E, gr.dL_dmeans2D, gr.dL_dconic, gr.dL_dcolors,           \
      gr.dL_dopacity, gr.dL_dmeans3D, gr.dL_dcov3D, gr.dL_dsh, gr.dL_dsh_rest, gr.dL_dscales, gr.dL_drotations,           \
      gr.dL_dextras, gr.dL_dsh_factors, in.tanfov_device, in.live_count, gr.stat_xyz_gradient_accum, gr.stat_denom,        \
      gr.stat_max_radii2D, (gr.stat_grad_multiplier != 0.f ? gr.stat_grad_multiplier : 1.0f), dbj, srj
  const int job = gr.deform_backward_job ? 1 : gr.sp_skinning_job ? (gr.sp_skinning_job->F == 8 ? 2 : 3) : 0;
#define SKGS_PB(JOB_)                                                                                  \
  if (in.colmap)                                                                                       \
    hipLaunchKernelGGL((preprocess_backward_kernel<true, JOB_>), grid, block, lds, s, SKGS_PB_ARGS);   \
  else                                                                                                 \
    hipLaunchKernelGGL((preprocess_backward_kernel<false, JOB_>), grid, block, lds, s, SKGS_PB_ARGS)
  switch (job) {
    case 1: SKGS_PB(1); break;
    case 2: SKGS_PB(2); break;
    case 3: SKGS_PB(3); break;
    default: SKGS_PB(0); break;
  }
#undef SKGS_PB
#undef SKGS_PB_ARGS
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sh_grad_from_factors(int P, int n_views, int D, int M, const float* factors, float* dL_dsh, float* dL_dsh_rest,
    hipStream_t s) {
  if (P == 0) return 0;
  if (M < 1 || M > 16 || D < 0 || (D + 1) * (D + 1) > M) return set_error("sh_grad_from_factors: bad degree %d / coefficient count %d", D, M);
  const int RL     = dL_dsh_rest ? (M - 1) * 3 : M * 3;
  const size_t lds = (size_t) PRE_THREADS * (RL | 1) * 4;
  hipLaunchKernelGGL(sh_grad_from_factors_kernel, dim3((P + PRE_THREADS - 1) / PRE_THREADS), dim3(PRE_THREADS), lds, s, P,
      n_views, D, M, factors, dL_dsh, dL_dsh_rest);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_mark_visible(int P, const float* means, const float* view, int colmap, uint8_t* present, hipStream_t s) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means, view, colmap, present);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace skgs
