/*
 * skgs_oracle.c -- CPU restatement of the SK_GS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity ORACLE.  It restates, in plain C, the arithmetic of the reference's
 * CUDA rasterizer and of the skeleton LBS deform, one function per reference kernel, in the
 * reference's own evaluation order.  It is NOT part of the product: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product (sk_gs_amd/) never does.
 *
 * Reference files followed (paths relative to the reference root, my_ext/_C/ = "C/"):
 *   C/src/nerf/gaussian_preprocess_colmap.cu   (colmap=1: upstream 3DGS arithmetic, column-major mats)
 *   C/src/nerf/gaussian_preprocess.cu          (colmap=0: row-major variant, restated bug-compatibly)
 *   C/src/nerf/gaussian_rasterizer_forward.cu  (SH->RGB, key duplication, tile ranges)
 *   C/src/nerf/gaussian_render.cu              (alpha-blend forward / backward)
 *   C/src/nerf/gaussian_rasterizer_backwrad.cu (SH backward, backward orchestration)
 *   C/src/nerf/gaussian_rasterizer_extra.cu    (extra feature blend fwd/bwd)
 *   C/src/nerf/gaussian_topk.cu                (per-pixel top-k weights)
 *   C/include/gaussian_render.h, C/include/ops_3d.h, C/include/lie.h
 *   networks/sk_gs.py:751-774,1143-1150,1192-1203 (LBS deform + activation epilogue)
 *
 * Parity pinning: the reference has no tests, fixtures or CPU path for the rasterizer and cannot be
 * compiled here (CUDA only).  This oracle is pinned instead by (a) the reference's own pure-torch
 * helpers imported in the build container (tests/golden/make_golden.py -> tests/golden/ npz files),
 * (b) an fp64 twin of this very file (compile with -DSKGS_F64) checked by finite differences, and
 * (c) closed-form cases.  Third-party pieces absent from the reference tree (lietorch, pytorch3d.knn,
 * diff_gaussian_rasterization) are "parity unpinned" beyond that -- see DESIGN.md.
 *
 * Build: see oracle/Makefile.  Compile with -ffp-contract=off so that a*b+c is never fused and the
 * evaluation order written here is the evaluation order executed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef SKGS_F64
typedef double REAL;
#define ORACLE(name) skgs_oracle64_##name
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#else
typedef float REAL;
#define ORACLE(name) skgs_oracle_##name
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#endif

#define RC(x) ((REAL) (x))
#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE 256
#define NUM_CHANNELS 3

/* exp used by the blend kernels.  exp_mode 0: libm (expf for the fp32 build) -- the literal restatement.
 * exp_mode 1 (fp32 build only): a reproducible exp made of IEEE double multiplies/adds, one operation per
 * statement; the HIP library's strict-math build (skgs_set_strict_math(1)) evaluates the very same sequence, so the
 * two sides can be compared bit for bit, threshold decisions included. */
static int g_exp_mode = 0;
void ORACLE(set_exp_mode)(int mode) { g_exp_mode = mode; }
static inline REAL oexp(REAL x) {
#ifndef SKGS_F64
  if (g_exp_mode == 1) {
    const double xd = (double) x;
    const double t  = xd * 1.4426950408889634;
    const double n  = rint(t);
    const double a  = n * 0.6931471803691238;
    const double b  = n * 1.9082149292705877e-10;
    double r        = xd - a;
    r               = r - b;
    double p        = r * (1.0 / 5040.0);
    p               = p + (1.0 / 720.0);
    p               = p * r;
    p               = p + (1.0 / 120.0);
    p               = p * r;
    p               = p + (1.0 / 24.0);
    p               = p * r;
    p               = p + (1.0 / 6.0);
    p               = p * r;
    p               = p + 0.5;
    p               = p * r;
    p               = p + 1.0;
    p               = p * r;
    p               = p + 1.0;
    return (float) ldexp(p, (int) n);
  }
#endif
  return R_EXP(x);
}

static inline REAL r_min(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL r_max(REAL a, REAL b) { return a > b ? a : b; }
static inline int i_min(int a, int b) { return a < b ? a : b; }
static inline int i_max(int a, int b) { return a > b ? a : b; }

/* ---- SH constants: C/include/gaussian_render.h:35-40 ---- */
static const REAL SH_C0   = RC(0.28209479177387814f);
static const REAL SH_C1   = RC(0.4886025119029199f);
static const REAL SH_C2[] = {RC(1.0925484305920792f), RC(-1.0925484305920792f), RC(0.31539156525252005f),
    RC(-1.0925484305920792f), RC(0.5462742152960396f)};
static const REAL SH_C3[] = {RC(-0.5900435899266435f), RC(2.890611442640554f), RC(-0.4570457994644658f),
    RC(0.3731763325901154f), RC(-0.4570457994644658f), RC(1.445305721320277f), RC(-0.5900435899266435f)};

/* ---- glm-style column-major 3x3 (m[col][row]), product order of glm type_mat3x3.inl:486-519 ---- */
typedef struct {
  REAL m[3][3];
} gmat3;

static gmat3 gmat3_make(REAL a0, REAL a1, REAL a2, REAL b0, REAL b1, REAL b2, REAL c0, REAL c1, REAL c2) {
  gmat3 r;
  r.m[0][0] = a0, r.m[0][1] = a1, r.m[0][2] = a2;
  r.m[1][0] = b0, r.m[1][1] = b1, r.m[1][2] = b2;
  r.m[2][0] = c0, r.m[2][1] = c1, r.m[2][2] = c2;
  return r;
}
static gmat3 gmat3_mul(const gmat3 a, const gmat3 b) {
  gmat3 r;
  for (int c = 0; c < 3; ++c)
    for (int w = 0; w < 3; ++w) r.m[c][w] = a.m[0][w] * b.m[c][0] + a.m[1][w] * b.m[c][1] + a.m[2][w] * b.m[c][2];
  return r;
}
static gmat3 gmat3_transpose(const gmat3 a) {
  gmat3 r;
  for (int c = 0; c < 3; ++c)
    for (int w = 0; w < 3; ++w) r.m[c][w] = a.m[w][c];
  return r;
}

/* ---- row-major helpers: C/include/ops_3d.h:38-63,93-103 ---- */
static void matmul_3x3x3(const REAL* A, const REAL* B, REAL* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) C[i * 3 + j] += A[i * 3 + k] * B[k * 3 + j];
}
static void matmul_3x3x3_tn(const REAL* At, const REAL* B, REAL* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) C[i * 3 + j] += At[k * 3 + i] * B[k * 3 + j];
}
static void quaternion_to_R(const REAL* q /*xyzw*/, REAL* R) {
  const REAL x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - z * w);
  R[2] = 2 * (y * w + x * z);
  R[3] = 2 * (x * y + z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);
  R[7] = 2 * (x * w + y * z);
  R[8] = 1 - 2 * (x * x + y * y);
}
/* C/include/ops_3d.h:118-126 */
static void dL_quaternion_to_R(const REAL* q, const REAL* dR, REAL* dq) {
  const REAL x = q[0], y = q[1], z = q[2], w = q[3];
  dq[0] = 2 * (-2 * x * (dR[4] + dR[8]) + y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) + w * (dR[7] - dR[5]));
  dq[1] = 2 * (x * (dR[1] + dR[3]) - 2 * y * (dR[0] + dR[8]) + z * (dR[5] + dR[7]) + w * (dR[2] - dR[6]));
  dq[2] = 2 * (x * (dR[2] + dR[6]) + y * (dR[5] + dR[7]) - 2 * z * (dR[0] + dR[4]) + w * (dR[3] - dR[1]));
  dq[3] = 2 * (x * (dR[7] - dR[5]) + y * (dR[2] - dR[6]) + z * (dR[3] - dR[1]));
}

/* ndc2Pix: the reference writes ((v + 1.0) * S - 1.0) * 0.5 with DOUBLE literals, so the float path is
 * evaluated in double and rounded once on return. gaussian_preprocess.cu:15, gaussian_preprocess_colmap.cu:26 */
static inline REAL ndc2Pix(REAL v, int S) { return (REAL) ((((double) v + 1.0) * S - 1.0) * 0.5); }

/* getRect: C/include/gaussian_render.h:42-47. max_radius is an int parameter in the reference. */
static void getRect(REAL px, REAL py, int max_radius, int gx, int gy, uint32_t* rmin, uint32_t* rmax) {
  rmin[0] = (uint32_t) i_min(gx, i_max(0, (int) ((px - max_radius) / BLOCK_X)));
  rmin[1] = (uint32_t) i_min(gy, i_max(0, (int) ((py - max_radius) / BLOCK_Y)));
  rmax[0] = (uint32_t) i_min(gx, i_max(0, (int) ((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
  rmax[1] = (uint32_t) i_min(gy, i_max(0, (int) ((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* dnormvdv(float3): C/include/gaussian_render.h:56-65 */
static void dnormvdv3(const REAL* v, const REAL* dv, REAL* out) {
  REAL sum2     = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  REAL invsum32 = RC(1.0) / R_SQRT(sum2 * sum2 * sum2);
  out[0]        = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  out[1]        = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  out[2]        = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* ================================================================================================
 * SH -> RGB forward: gaussian_rasterizer_forward.cu:97-137
 * ============================================================================================== */
static void computeColorFromSH_fwd(int idx, int deg, int max_coeffs, const REAL* means, const REAL* campos,
    const REAL* shs, uint8_t* clamped, REAL* out) {
  const REAL* pos = means + 3 * idx;
  REAL dir[3]     = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  REAL len        = R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); /* glm::length = sqrt(dot) */
  dir[0] = dir[0] / len, dir[1] = dir[1] / len, dir[2] = dir[2] / len;
  const REAL* sh = shs + (size_t) idx * max_coeffs * 3;
  REAL res[3];
  for (int c = 0; c < 3; ++c) {
#define SH(i) sh[(i) *3 + c]
    REAL r = SH_C0 * SH(0);
    if (deg > 0) {
      REAL x = dir[0], y = dir[1], z = dir[2];
      r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
      if (deg > 1) {
        REAL xx = x * x, yy = y * y, zz = z * z;
        REAL xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (RC(2.0) * zz - xx - yy) * SH(6) +
            SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          r = r + SH_C3[0] * y * (RC(3.0) * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
              SH_C3[2] * y * (RC(4.0) * zz - xx - yy) * SH(11) +
              SH_C3[3] * z * (RC(2.0) * zz - RC(3.0) * xx - RC(3.0) * yy) * SH(12) +
              SH_C3[4] * x * (RC(4.0) * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
              SH_C3[6] * x * (xx - RC(3.0) * yy) * SH(15);
        }
      }
    }
#undef SH
    r += RC(0.5);
    clamped[3 * idx + c] = (r < 0);
    res[c]               = r_max(r, RC(0.0));
  }
  out[0] = res[0], out[1] = res[1], out[2] = res[2];
}

/* ================================================================================================
 * SH -> RGB backward: gaussian_rasterizer_backwrad.cu:26-127.  dL_dmeans is ACCUMULATED (+=).
 * ============================================================================================== */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const REAL* means, const REAL* campos,
    const REAL* shs, const uint8_t* clamped, const REAL* dL_dcolor, REAL* dL_dmeans, REAL* dL_dshs) {
  const REAL* pos  = means + 3 * idx;
  REAL dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  REAL len         = R_SQRT(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  REAL x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
  const REAL* sh = shs + (size_t) idx * max_coeffs * 3;
  REAL* dL_dsh   = dL_dshs + (size_t) idx * max_coeffs * 3;
  REAL dL_dRGB[3];
  for (int c = 0; c < 3; ++c) dL_dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? RC(0.0) : RC(1.0));
  REAL dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SHV(i, c) sh[(i) *3 + (c)]
#define SETSH(i, coef)                                                  \
  for (int c = 0; c < 3; ++c) dL_dsh[(i) *3 + c] = (coef) * dL_dRGB[c];
  SETSH(0, SH_C0);
  if (deg > 0) {
    REAL dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
    SETSH(1, dRGBdsh1);
    SETSH(2, dRGBdsh2);
    SETSH(3, dRGBdsh3);
    for (int c = 0; c < 3; ++c) {
      dRGBdx[c] = -SH_C1 * SHV(3, c);
      dRGBdy[c] = -SH_C1 * SHV(1, c);
      dRGBdz[c] = SH_C1 * SHV(2, c);
    }
    if (deg > 1) {
      REAL xx = x * x, yy = y * y, zz = z * z;
      REAL xy = x * y, yz = y * z, xz = x * z;
      REAL dRGBdsh4 = SH_C2[0] * xy, dRGBdsh5 = SH_C2[1] * yz, dRGBdsh6 = SH_C2[2] * (RC(2.0) * zz - xx - yy);
      REAL dRGBdsh7 = SH_C2[3] * xz, dRGBdsh8 = SH_C2[4] * (xx - yy);
      SETSH(4, dRGBdsh4);
      SETSH(5, dRGBdsh5);
      SETSH(6, dRGBdsh6);
      SETSH(7, dRGBdsh7);
      SETSH(8, dRGBdsh8);
      for (int c = 0; c < 3; ++c) {
        dRGBdx[c] += SH_C2[0] * y * SHV(4, c) + SH_C2[2] * RC(2.0) * -x * SHV(6, c) + SH_C2[3] * z * SHV(7, c) +
                     SH_C2[4] * RC(2.0) * x * SHV(8, c);
        dRGBdy[c] += SH_C2[0] * x * SHV(4, c) + SH_C2[1] * z * SHV(5, c) + SH_C2[2] * RC(2.0) * -y * SHV(6, c) +
                     SH_C2[4] * RC(2.0) * -y * SHV(8, c);
        dRGBdz[c] += SH_C2[1] * y * SHV(5, c) + SH_C2[2] * RC(2.0) * RC(2.0) * z * SHV(6, c) + SH_C2[3] * x * SHV(7, c);
      }
      if (deg > 2) {
        REAL dRGBdsh9  = SH_C3[0] * y * (RC(3.0) * xx - yy);
        REAL dRGBdsh10 = SH_C3[1] * xy * z;
        REAL dRGBdsh11 = SH_C3[2] * y * (RC(4.0) * zz - xx - yy);
        REAL dRGBdsh12 = SH_C3[3] * z * (RC(2.0) * zz - RC(3.0) * xx - RC(3.0) * yy);
        REAL dRGBdsh13 = SH_C3[4] * x * (RC(4.0) * zz - xx - yy);
        REAL dRGBdsh14 = SH_C3[5] * z * (xx - yy);
        REAL dRGBdsh15 = SH_C3[6] * x * (xx - RC(3.0) * yy);
        SETSH(9, dRGBdsh9);
        SETSH(10, dRGBdsh10);
        SETSH(11, dRGBdsh11);
        SETSH(12, dRGBdsh12);
        SETSH(13, dRGBdsh13);
        SETSH(14, dRGBdsh14);
        SETSH(15, dRGBdsh15);
        for (int c = 0; c < 3; ++c) {
          dRGBdx[c] += (SH_C3[0] * SHV(9, c) * RC(3.0) * RC(2.0) * xy + SH_C3[1] * SHV(10, c) * yz +
                        SH_C3[2] * SHV(11, c) * RC(-2.0) * xy + SH_C3[3] * SHV(12, c) * RC(-3.0) * RC(2.0) * xz +
                        SH_C3[4] * SHV(13, c) * (RC(-3.0) * xx + RC(4.0) * zz - yy) +
                        SH_C3[5] * SHV(14, c) * RC(2.0) * xz + SH_C3[6] * SHV(15, c) * RC(3.0) * (xx - yy));
          dRGBdy[c] += (SH_C3[0] * SHV(9, c) * RC(3.0) * (xx - yy) + SH_C3[1] * SHV(10, c) * xz +
                        SH_C3[2] * SHV(11, c) * (RC(-3.0) * yy + RC(4.0) * zz - xx) +
                        SH_C3[3] * SHV(12, c) * RC(-3.0) * RC(2.0) * yz + SH_C3[4] * SHV(13, c) * RC(-2.0) * xy +
                        SH_C3[5] * SHV(14, c) * RC(-2.0) * yz + SH_C3[6] * SHV(15, c) * RC(-3.0) * RC(2.0) * xy);
          dRGBdz[c] += (SH_C3[1] * SHV(10, c) * xy + SH_C3[2] * SHV(11, c) * RC(4.0) * RC(2.0) * yz +
                        SH_C3[3] * SHV(12, c) * RC(3.0) * (RC(2.0) * zz - xx - yy) +
                        SH_C3[4] * SHV(13, c) * RC(4.0) * RC(2.0) * xz + SH_C3[5] * SHV(14, c) * (xx - yy));
        }
      }
    }
  }
#undef SHV
#undef SETSH
  /* glm::dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z */
  REAL dL_ddir[3] = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
      dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
      dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
  REAL dL_dmean[3];
  dnormvdv3(dir_orig, dL_ddir, dL_dmean);
  dL_dmeans[3 * idx + 0] += dL_dmean[0];
  dL_dmeans[3 * idx + 1] += dL_dmean[1];
  dL_dmeans[3 * idx + 2] += dL_dmean[2];
}

/* ================================================================================================
 * colmap = 1 : gaussian_preprocess_colmap.cu
 * ============================================================================================== */
static inline void transformPoint4x3_colmap(const REAL* p, const REAL* m, REAL* o) { /* :28-35 */
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void transformPoint4x4_colmap(const REAL* p, const REAL* m, REAL* o) { /* :37-43 */
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline void transformVec4x3Transpose_colmap(const REAL* p, const REAL* m, REAL* o) { /* :54-61 */
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

/* computeCov3D_colmap fwd: gaussian_preprocess_colmap.cu:121-152 (quaternion xyzw, r = q.w, NOT normalised) */
static gmat3 colmap_R(const REAL* rot) {
  REAL x = rot[0], y = rot[1], z = rot[2], r = rot[3];
  return gmat3_make(RC(1.) - RC(2.) * (y * y + z * z), RC(2.) * (x * y - r * z), RC(2.) * (x * z + r * y),
      RC(2.) * (x * y + r * z), RC(1.) - RC(2.) * (x * x + z * z), RC(2.) * (y * z - r * x),
      RC(2.) * (x * z - r * y), RC(2.) * (y * z + r * x), RC(1.) - RC(2.) * (x * x + y * y));
}
static void computeCov3D_colmap(const REAL* scale, REAL mod, const REAL* rot, REAL* cov3D) {
  gmat3 S     = gmat3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0]   = mod * scale[0];
  S.m[1][1]   = mod * scale[1];
  S.m[2][2]   = mod * scale[2];
  gmat3 R     = colmap_R(rot);
  gmat3 M     = gmat3_mul(S, R);
  gmat3 Sigma = gmat3_mul(gmat3_transpose(M), M);
  cov3D[0]    = Sigma.m[0][0];
  cov3D[1]    = Sigma.m[0][1];
  cov3D[2]    = Sigma.m[0][2];
  cov3D[3]    = Sigma.m[1][1];
  cov3D[4]    = Sigma.m[1][2];
  cov3D[5]    = Sigma.m[2][2];
}

/* computeCov2D_colmap: gaussian_preprocess_colmap.cu:85-116 */
static void computeCov2D_colmap(const REAL* mean, REAL focal_x, REAL focal_y, REAL tan_fovx, REAL tan_fovy,
    const REAL* cov3D, const REAL* vm, REAL* cov /*3*/) {
  REAL t[3];
  transformPoint4x3_colmap(mean, vm, t);
  const REAL limx = RC(1.3f) * tan_fovx;
  const REAL limy = RC(1.3f) * tan_fovy;
  const REAL txtz = t[0] / t[2];
  const REAL tytz = t[1] / t[2];
  t[0]            = r_min(limx, r_max(-limx, txtz)) * t[2];
  t[1]            = r_min(limy, r_max(-limy, tytz)) * t[2];
  gmat3 J         = gmat3_make(focal_x / t[2], 0, -(focal_x * t[0]) / (t[2] * t[2]), 0, focal_y / t[2],
              -(focal_y * t[1]) / (t[2] * t[2]), 0, 0, 0);
  gmat3 W         = gmat3_make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  gmat3 T         = gmat3_mul(W, J);
  gmat3 Vrk = gmat3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  gmat3 c   = gmat3_mul(gmat3_mul(gmat3_transpose(T), gmat3_transpose(Vrk)), T);
  c.m[0][0] += RC(0.3f);
  c.m[1][1] += RC(0.3f);
  cov[0] = c.m[0][0], cov[1] = c.m[0][1], cov[2] = c.m[1][1];
}

/* ================================================================================================
 * colmap = 0 : gaussian_preprocess.cu (row-major)
 * ============================================================================================== */
static inline void xfm_p_4x3(const REAL* p, const REAL* m, REAL* o) { /* ops_3d.h:136-144 */
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
}
static inline void xfm_p_4x4(const REAL* p, const REAL* m, REAL* o) { /* ops_3d.h:146-153 */
  o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
  o[3] = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
}
static inline void xfm_v_4x3_T(const REAL* p, const REAL* m, REAL* o) { /* ops_3d.h:165-173 */
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2];
}
/* computeCov3D: gaussian_preprocess.cu:79-96 (scale_modifier NOT applied) */
static void computeCov3D_rowmajor(const REAL* s, const REAL* rot, REAL* cov3D) {
  REAL R[9] = {0};
  quaternion_to_R(rot, R);
  REAL sx2 = s[0] * s[0], sy2 = s[1] * s[1], sz2 = s[2] * s[2];
  cov3D[0] = R[0] * R[0] * sx2 + R[1] * R[1] * sy2 + R[2] * R[2] * sz2;
  cov3D[1] = R[0] * R[3] * sx2 + R[1] * R[4] * sy2 + R[2] * R[5] * sz2;
  cov3D[2] = R[0] * R[6] * sx2 + R[1] * R[7] * sy2 + R[2] * R[8] * sz2;
  cov3D[3] = R[3] * R[3] * sx2 + R[4] * R[4] * sy2 + R[5] * R[5] * sz2;
  cov3D[4] = R[3] * R[6] * sx2 + R[4] * R[7] * sy2 + R[5] * R[8] * sz2;
  cov3D[5] = R[6] * R[6] * sx2 + R[7] * R[7] * sy2 + R[8] * R[8] * sz2;
}
/* computeCov2D: gaussian_preprocess.cu:39-75.  NOTE forms M = W*J on row-major arrays and Sigma2D = M^T V M,
 * which is NOT the colmap arithmetic; restated literally. */
static void computeCov2D_rowmajor(const REAL* mean, REAL focal_x, REAL focal_y, REAL tan_fovx, REAL tan_fovy,
    const REAL* cov3D, const REAL* vm, REAL* cov /*3*/) {
  REAL t[3];
  xfm_p_4x3(mean, vm, t);
  const REAL limx = RC(1.3f) * tan_fovx;
  const REAL limy = RC(1.3f) * tan_fovy;
  const REAL txtz = t[0] / t[2];
  const REAL tytz = t[1] / t[2];
  t[0]            = r_min(r_max(txtz, -limx), limx) * t[2]; /* clamp(): util.cuh:62-64 */
  t[1]            = r_min(r_max(tytz, -limy), limy) * t[2];
  REAL J[9] = {focal_x / t[2], 0, -(focal_x * t[0]) / (t[2] * t[2]), 0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
      0, 0, 0};
  REAL W[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  REAL M[9] = {0};
  matmul_3x3x3(W, J, M);
  REAL Vrk[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
  REAL tmp[9] = {0};
  matmul_3x3x3_tn(M, Vrk, tmp);
  REAL c[9] = {0};
  matmul_3x3x3(tmp, M, c);
  c[0] += RC(0.3f);
  c[4] += RC(0.3f);
  cov[0] = c[0], cov[1] = c[1], cov[2] = c[4];
}

/* ================================================================================================
 * preprocessCUDA / preprocessCUDA_colmap: gaussian_preprocess.cu:99-168, gaussian_preprocess_colmap.cu:155-224
 * One iteration per Gaussian.  Culled Gaussians leave means2D/depths/conic/rgb untouched (stale), as in the
 * reference; radii and tiles_touched are zeroed first for every Gaussian.
 * ============================================================================================== */
void ORACLE(preprocess_forward)(int P, int D, int M, const REAL* means3D, const REAL* scales, REAL scale_modifier,
    const REAL* rotations, const REAL* opacities, const REAL* shs, const REAL* cov3D_precomp,
    const REAL* colors_precomp, const REAL* viewmatrix, const REAL* projmatrix, const REAL* cam_pos, int W, int H,
    REAL tan_fovx, REAL tan_fovy, int colmap, int32_t* radii, REAL* means2D, REAL* depths, REAL* cov3Ds, REAL* rgb,
    REAL* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped) {
  /* gaussian_rasterizer_forward.cu:163-164 */
  const REAL focal_y = H / (RC(2.0) * tan_fovy);
  const REAL focal_x = W / (RC(2.0) * tan_fovx);
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    radii[idx]         = 0;
    tiles_touched[idx] = 0;
    const REAL* p_orig = means3D + 3 * idx;
    REAL p_view[3];
    REAL p_hom[4];
    if (colmap) {
      transformPoint4x3_colmap(p_orig, viewmatrix, p_view);
      if (p_view[2] <= RC(0.2f)) continue; /* in_frustum_colmap :73 */
      transformPoint4x4_colmap(p_orig, projmatrix, p_hom);
    } else {
      xfm_p_4x3(p_orig, viewmatrix, p_view);
      if (p_view[2] <= RC(-1.0f)) continue; /* in_frustum :28 */
      xfm_p_4x4(p_orig, projmatrix, p_hom);
    }
    REAL p_w       = RC(1.0) / (p_hom[3] + RC(0.0000001f));
    REAL p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
    const REAL* cov3D;
    if (cov3D_precomp != NULL) {
      cov3D = cov3D_precomp + 6 * idx;
    } else {
      if (colmap)
        computeCov3D_colmap(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
      else
        computeCov3D_rowmajor(scales + 3 * idx, rotations + 4 * idx, cov3Ds + 6 * idx);
      cov3D = cov3Ds + 6 * idx;
    }
    REAL cov[3];
    if (colmap)
      computeCov2D_colmap(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
    else
      computeCov2D_rowmajor(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
    REAL det = (cov[0] * cov[2] - cov[1] * cov[1]);
    if (det == RC(0.0)) continue;
    REAL det_inv   = RC(1.) / det;
    REAL conic[3]  = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
    REAL mid       = RC(0.5) * (cov[0] + cov[2]);
    REAL lambda1   = mid + R_SQRT(r_max(RC(0.1f), mid * mid - det));
    REAL lambda2   = mid - R_SQRT(r_max(RC(0.1f), mid * mid - det));
    REAL my_radius = R_CEIL(RC(3.) * R_SQRT(r_max(lambda1, lambda2)));
    REAL pix[2]    = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
    uint32_t rmin[2], rmax[2];
    getRect(pix[0], pix[1], (int) my_radius, gx, gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
    if (colors_precomp == NULL) {
      REAL res[3];
      computeColorFromSH_fwd(idx, D, M, means3D, cam_pos, shs, clamped, res);
      rgb[3 * idx + 0] = res[0], rgb[3 * idx + 1] = res[1], rgb[3 * idx + 2] = res[2];
    }
    depths[idx]                = p_view[2];
    radii[idx]                 = (int32_t) my_radius;
    means2D[2 * idx + 0]       = pix[0];
    means2D[2 * idx + 1]       = pix[1];
    conic_opacity[4 * idx + 0] = conic[0];
    conic_opacity[4 * idx + 1] = conic[1];
    conic_opacity[4 * idx + 2] = conic[2];
    conic_opacity[4 * idx + 3] = opacities[idx];
    tiles_touched[idx]         = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
  }
}

/* getHigherMsb: gaussian_rasterizer_forward.cu:30-42 */
uint32_t ORACLE(getHigherMsb)(uint32_t n) {
  uint32_t msb  = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb)
      msb += step;
    else
      msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

typedef struct {
  uint64_t key;
  uint32_t val;
  uint32_t seq; /* emission order: makes the comparator a stable sort, as CUB radix sort is */
} kv_t;
static int kv_cmp(const void* a, const void* b) {
  const kv_t* x = (const kv_t*) a;
  const kv_t* y = (const kv_t*) b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
  return 0;
}

/* InclusiveSum + duplicateWithKeys + SortPairs + identifyTileRanges:
 * gaussian_rasterizer_forward.cu:45-94,203-241.  Returns num_rendered.  keys/point_list need capacity >= R
 * (call once with capacity 0 to obtain R).  ranges is [tiles][2], zero-filled here. */
int64_t ORACLE(bin_and_sort)(int P, int W, int H, const REAL* means2D, const REAL* depths, const int32_t* radii,
    const uint32_t* tiles_touched, uint32_t* point_offsets, int64_t capacity, uint64_t* point_list_keys,
    uint32_t* point_list, uint32_t* ranges) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  uint32_t acc = 0;
  for (int i = 0; i < P; ++i) { /* cub::DeviceScan::InclusiveSum */
    acc += tiles_touched[i];
    point_offsets[i] = acc;
  }
  const int64_t R = P > 0 ? (int64_t) point_offsets[P - 1] : 0;
  memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t) gx * gy);
  if (R > capacity || R == 0) return R;
  kv_t* kv = (kv_t*) malloc(sizeof(kv_t) * (size_t) R);
  for (int idx = 0; idx < P; ++idx) { /* duplicateWithKeys :45-73 */
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
      uint32_t rmin[2], rmax[2];
      getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
      float depth_f = (float) depths[idx];
      uint32_t dbits;
      memcpy(&dbits, &depth_f, 4);
      for (int y = (int) rmin[1]; y < (int) rmax[1]; y++) {
        for (int x = (int) rmin[0]; x < (int) rmax[0]; x++) {
          uint64_t key = (uint64_t) (y * gx + x);
          key <<= 32;
          key |= dbits;
          kv[off].key = key;
          kv[off].val = (uint32_t) idx;
          kv[off].seq = off;
          off++;
        }
      }
    }
  }
  /* SortPairs over bits [0, 32+getHigherMsb(tiles)): tile ids are < 2^msb, so this is a full-key stable sort.
   * The same order -- (tile, depth bits, emission order) -- is produced in two stable steps: a counting sort by the
   * tile id (the key's high word), then every tile's span sorted by (key, emission order) on its own, the spans in
   * parallel (the CPU baseline of bench.py times this oracle: one serial qsort of R entries would be most of it) */
  {
    const int T    = gx * gy;
    uint32_t* head = (uint32_t*) calloc((size_t) T + 1, sizeof(uint32_t));
    for (int64_t i = 0; i < R; ++i) head[(kv[i].key >> 32) + 1]++;
    for (int t = 0; t < T; ++t) head[t + 1] += head[t];
    kv_t* by_tile   = (kv_t*) malloc(sizeof(kv_t) * (size_t) R);
    uint32_t* fill = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) T);
    memcpy(fill, head, sizeof(uint32_t) * (size_t) T);
    for (int64_t i = 0; i < R; ++i) by_tile[fill[kv[i].key >> 32]++] = kv[i];
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < T; ++t)
      if (head[t + 1] - head[t] > 1) qsort(by_tile + head[t], head[t + 1] - head[t], sizeof(kv_t), kv_cmp);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < R; ++i) {
      point_list_keys[i] = by_tile[i].key;
      point_list[i]      = by_tile[i].val;
    }
    free(fill);
    free(by_tile);
    free(head);
  }
  free(kv);
  for (int64_t idx = 0; idx < R; ++idx) { /* identifyTileRanges :77-94 */
    uint32_t currtile = (uint32_t) (point_list_keys[idx] >> 32);
    if (idx == 0)
      ranges[2 * currtile + 0] = 0;
    else {
      uint32_t prevtile = (uint32_t) (point_list_keys[idx - 1] >> 32);
      if (currtile != prevtile) {
        ranges[2 * prevtile + 1] = (uint32_t) idx;
        ranges[2 * currtile + 0] = (uint32_t) idx;
      }
    }
    if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t) R;
  }
  return R;
}

/* ================================================================================================
 * renderCUDA_forward<3,E>: gaussian_render.cu:16-112.  One iteration per pixel; the block/batch structure of
 * the CUDA kernel does not change any per-pixel result, so the pixel loop walks its tile's list directly.
 * ============================================================================================== */
void ORACLE(render_forward)(int W, int H, int E, const uint32_t* ranges, const uint32_t* point_list,
    const REAL* means2D, const REAL* features, const REAL* conic_opacity, const REAL* extra, uint32_t* n_contrib,
    REAL* out_color, REAL* out_opacity, REAL* out_extra) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        REAL T                = RC(1.0);
        uint32_t contributor = 0, last_contributor = 0;
        REAL C[NUM_CHANNELS + 16] = {0};
        REAL* Ex                  = C + NUM_CHANNELS;
        for (uint32_t k = r0; k < r1; ++k) {
          contributor++;
          const uint32_t id = point_list[k];
          REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const REAL* con_o = conic_opacity + 4 * id;
          REAL power        = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
          if (power > RC(0.0)) continue;
          REAL alpha = r_min(RC(0.99f), con_o[3] * oexp(power));
          if (alpha < RC(1.0f / 255.0f)) continue;
          REAL test_T = T * (1 - alpha);
          if (test_T < RC(0.0001f)) break; /* done = true */
          for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
          for (int ch = 0; ch < E; ch++) Ex[ch] += extra[id * E + ch] * alpha * T;
          T                = test_T;
          last_contributor = contributor;
        }
        out_opacity[pix_id] = RC(1.) - T;
        n_contrib[pix_id]   = last_contributor;
        for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[(size_t) ch * H * W + pix_id] = C[ch];
        for (int ch = 0; ch < E; ch++) out_extra[(size_t) ch * H * W + pix_id] = Ex[ch];
      }
  }
}

/* ================================================================================================
 * Flip census (test infrastructure of the parity tests, no reference counterpart): how close does each pixel's walk
 * of renderCUDA_forward (gaussian_render.cu:78-100) come to one of its three data-dependent branches, in units of
 * the rounding error an implementation can have there?  With t = |A dx^2|/2 + |C dy^2|/2 + |B dx dy| (the magnitude
 * of the terms `power` is summed from: its absolute rounding error is a few ulp OF t, and that is also the relative
 * error of exp(power)):
 *   power > 0            margin |power| / t
 *   alpha < 1/255        margin |255 o exp(power) - 1| / (1 + t)
 *   T (1 - alpha) < 1e-4 margin |T (1 - alpha) / 1e-4 - 1| / sum over the splats blended so far of
 *                        (1 + (1 + t_j) alpha_j / (1 - alpha_j))   (T carries the error of every alpha before it)
 * Two implementations that round exp / the quadratic form differently (CUDA expf with FMA contraction, libm expf
 * without, v_exp_f32 on a pre-scaled conic) take different branches exactly at pairs whose margin is a small
 * multiple of 2^-24 = 6e-8.  pix_margin[H*W] = the smallest margin of the pixel's walk.
 * ============================================================================================== */
void ORACLE(render_margins)(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const REAL* means2D,
    const REAL* conic_opacity, REAL* pix_margin, uint8_t* pix_cause /* optional: 0 power, 1 alpha, 2 T */) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        REAL T = RC(1.0), margin = RC(1e30);
        double t_err = 0.0; /* relative error of T in units of the rounding */
        uint8_t cause = 255;
        for (uint32_t k = r0; k < r1; ++k) {
          const uint32_t id = point_list[k];
          const REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const REAL* con_o = conic_opacity + 4 * id;
          const REAL power  = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
          const REAL terms  = RC(0.5) * (REAL) (fabs((double) (con_o[0] * dx * dx)) + fabs((double) (con_o[2] * dy * dy))) +
                             (REAL) fabs((double) (con_o[1] * dx * dy));
          if (terms > 0) {
            const REAL m = (REAL) fabs((double) power) / terms;
            if (m < margin) margin = m, cause = 0;
          }
          if (power > RC(0.0)) continue;
          const REAL araw = con_o[3] * oexp(power);
          {
            const REAL m = (REAL) fabs((double) (araw * RC(255.0) - RC(1.0))) / (RC(1.0) + terms);
            if (m < margin) margin = m, cause = 1;
          }
          const REAL alpha = r_min(RC(0.99f), araw);
          if (alpha < RC(1.0f / 255.0f)) continue;
          const REAL test_T = T * (1 - alpha);
          t_err += 1.0 + (1.0 + (double) terms) * (double) alpha / (1.0 - (double) alpha);
          {
            const REAL m = (REAL) (fabs((double) (test_T / RC(0.0001f) - RC(1.0))) / t_err);
            if (m < margin) margin = m, cause = 2;
          }
          if (test_T < RC(0.0001f)) break;
          T = test_T;
        }
        pix_margin[pix_id] = margin;
        if (pix_cause) pix_cause[pix_id] = cause;
      }
  }
}

/* The fingerprint the HIP census kernel writes (include/skgs.h, skgs_render_census), from the reference walk:
 * census[pix][0] = number of list entries blended, [1] = sum of mix(1-based list position), mix(k) =
 * (k * 2654435761) ^ (k >> 5) mod 2^32.  Two walks with equal fingerprints took the same branch at every entry (up to
 * a hash collision of 2^-32). */
void ORACLE(render_census)(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const REAL* means2D,
    const REAL* conic_opacity, uint32_t* census) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        REAL T                = RC(1.0);
        uint32_t contributor = 0, n = 0, h = 0;
        for (uint32_t k = r0; k < r1; ++k) {
          contributor++;
          const uint32_t id = point_list[k];
          const REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const REAL* con_o = conic_opacity + 4 * id;
          const REAL power  = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
          if (power > RC(0.0)) continue;
          const REAL alpha = r_min(RC(0.99f), con_o[3] * oexp(power));
          if (alpha < RC(1.0f / 255.0f)) continue;
          const REAL test_T = T * (1 - alpha);
          if (test_T < RC(0.0001f)) break;
          T = test_T;
          n += 1;
          h += (contributor * 2654435761u) ^ (contributor >> 5);
        }
        census[2 * (size_t) pix_id]     = n;
        census[2 * (size_t) pix_id + 1] = h;
      }
  }
}

/* gauss_flag[id] = 1 for every Gaussian that (nearly: alpha >= 0.99/255) contributes to a pixel of pix_mask -- the
 * rows of the per-Gaussian gradients a branch flip at one of those pixels can change (a flip moves T for everything
 * behind it and the behind-colour for everything in front) */
void ORACLE(render_touching)(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const REAL* means2D,
    const REAL* conic_opacity, const uint8_t* pix_mask, uint8_t* gauss_flag) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      if (!pix_mask[(size_t) W * py + px]) continue;
      const int tile    = (py / BLOCK_Y) * gx + px / BLOCK_X;
      const REAL pixf[2] = {(REAL) px, (REAL) py};
      for (uint32_t k = ranges[2 * tile]; k < ranges[2 * tile + 1]; ++k) {
        const uint32_t id = point_list[k];
        const REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
        const REAL* con_o = conic_opacity + 4 * id;
        const REAL power  = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
        if (power > RC(1e-3)) continue;
        if (con_o[3] * oexp(r_min(power, RC(0.0))) * RC(255.0) < RC(0.99)) continue;
        gauss_flag[id] = 1;
      }
    }
}

/* ================================================================================================
 * renderCUDA_backward<3,E>: gaussian_render.cu:182-341.  The CUDA kernel scatters with atomicAdd (order
 * undefined); here tiles are walked in parallel, every tile INSTANCE (entry k of the sorted list) owns one private
 * accumulator row filled by its tile's pixels in pixel order, and the rows are then added to their Gaussians in
 * list order k = 0..R-1: deterministic for any thread count, and no per-thread [P] image to allocate and reduce
 * (what made 128 cores slower than one).  dL_dmean2D is [P,3], dL_dconic2D is [P,4] (x,y,_,w), accumulated INTO.
 * ============================================================================================== */
#ifdef _OPENMP
#include <omp.h>
#endif
void ORACLE(render_backward)(int P, int W, int H, int E, const uint32_t* ranges, const uint32_t* point_list,
    const REAL* means2D, const REAL* conic_opacity, const REAL* colors, const REAL* extras, const REAL* out_opacity,
    const uint32_t* n_contrib, const REAL* dL_dpixels, const REAL* dL_dout_extra, const REAL* dL_dout_opacity,
    REAL* dL_dmean2D, REAL* dL_dconic2D, REAL* dL_dopacity, REAL* dL_dcolors, REAL* dL_dextras) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const int C  = NUM_CHANNELS;
  const int NV    = 9 + E; /* mean2D.x,.y conic.x,.y,.w opacity color[3] extra[E] */
  uint32_t R      = 0;
  for (int tile = 0; tile < gx * gy; ++tile)
    if (ranges[2 * tile + 1] > R) R = ranges[2 * tile + 1];
  REAL* scratch   = (REAL*) calloc((size_t) (R > 0 ? R : 1) * NV, sizeof(REAL));
  const REAL ddelx_dx = (REAL) (0.5 * W);
  const REAL ddely_dy = (REAL) (0.5 * H);
  {
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
      const int tx = tile % gx, ty = tile / gx;
      const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
      for (int ly = 0; ly < BLOCK_Y; ++ly)
        for (int lx = 0; lx < BLOCK_X; ++lx) {
          const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
          if (!(px < W && py < H)) continue;
          const uint32_t pix_id = (uint32_t) W * py + px;
          const REAL pixf[2]    = {(REAL) px, (REAL) py};
          const REAL T_final    = RC(1.0) - out_opacity[pix_id];
          REAL T                = T_final;
          const REAL dL_dT      = -dL_dout_opacity[pix_id];
          uint32_t contributor  = r1 - r0;
          const int last_contributor = (int) n_contrib[pix_id];
          REAL accum_rec[NUM_CHANNELS + 16] = {0};
          REAL dL_dpixel[NUM_CHANNELS + 16];
          REAL last_color[NUM_CHANNELS + 16] = {0};
          for (int c = 0; c < C; c++) dL_dpixel[c] = dL_dpixels[(size_t) c * H * W + pix_id];
          for (int e = 0; e < E; e++) dL_dpixel[C + e] = dL_dout_extra[(size_t) e * H * W + pix_id];
          REAL last_alpha = 0;
          for (uint32_t k = r1; k-- > r0;) {
            contributor--;
            if ((int64_t) contributor >= (int64_t) last_contributor) continue;
            const uint32_t id = point_list[k];
            const REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
            const REAL* con_o = conic_opacity + 4 * id;
            const REAL power  = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
            if (power > RC(0.0)) continue;
            const REAL G     = oexp(power);
            const REAL alpha = r_min(RC(0.99f), con_o[3] * G);
            if (alpha < RC(1.0f / 255.0f)) continue;
            T                          = T / (RC(1.) - alpha);
            const REAL dchannel_dcolor = alpha * T;
            REAL dL_dalpha             = RC(0.0);
            REAL* a                    = scratch + (size_t) k * NV;
            for (int ch = 0; ch < C; ch++) {
              const REAL c  = colors[id * C + ch];
              accum_rec[ch] = last_alpha * last_color[ch] + (RC(1.) - last_alpha) * accum_rec[ch];
              last_color[ch] = c;
              const REAL dL_dchannel = dL_dpixel[ch];
              dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
              a[6 + ch] += dchannel_dcolor * dL_dchannel;
            }
            for (int ch = 0; ch < E; ch++) {
              const REAL c      = extras[id * E + ch];
              accum_rec[C + ch] = last_alpha * last_color[C + ch] + (RC(1.) - last_alpha) * accum_rec[C + ch];
              last_color[C + ch] = c;
              const REAL dL_dchannel = dL_dpixel[C + ch];
              dL_dalpha += (c - accum_rec[C + ch]) * dL_dchannel;
              a[9 + ch] += dchannel_dcolor * dL_dchannel;
            }
            dL_dalpha *= T;
            last_alpha = alpha;
            dL_dalpha += (-T_final / (RC(1.) - alpha)) * dL_dT;
            const REAL dL_dG    = con_o[3] * dL_dalpha;
            const REAL gdx      = G * dx;
            const REAL gdy      = G * dy;
            const REAL dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
            const REAL dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
            a[0] += dL_dG * dG_ddelx * ddelx_dx;
            a[1] += dL_dG * dG_ddely * ddely_dy;
            a[2] += RC(-0.5) * gdx * dx * dL_dG;
            a[3] += RC(-0.5) * gdx * dy * dL_dG;
            a[4] += RC(-0.5) * gdy * dy * dL_dG;
            a[5] += G * dL_dalpha;
          }
        }
    }
  }
  (void) P;
  {
    for (uint32_t k = 0; k < R; ++k) {
      const uint32_t id = point_list[k];
      const REAL* a     = scratch + (size_t) k * NV;
      dL_dmean2D[3 * id + 0] += a[0];
      dL_dmean2D[3 * id + 1] += a[1];
      dL_dconic2D[4 * id + 0] += a[2];
      dL_dconic2D[4 * id + 1] += a[3];
      dL_dconic2D[4 * id + 3] += a[4];
      dL_dopacity[id] += a[5];
      for (int ch = 0; ch < C; ++ch) dL_dcolors[id * C + ch] += a[6 + ch];
      for (int ch = 0; ch < E; ++ch) dL_dextras[id * E + ch] += a[9 + ch];
    }
  }
  free(scratch);
}

/* ================================================================================================
 * computeCov2DCUDA / computeCov2DCUDA_colmap (backward of conic): gaussian_preprocess.cu:183-298,
 * gaussian_preprocess_colmap.cu:240-354.   dL_dmeans is ASSIGNED; dL_dcov written.
 * ============================================================================================== */
static void cov2D_backward_colmap(int idx, const REAL* means, const REAL* cov3Ds, REAL h_x, REAL h_y, REAL tan_fovx,
    REAL tan_fovy, const REAL* vm, const REAL* dL_dconics, REAL* dL_dmeans, REAL* dL_dcov) {
  const REAL* cov3D  = cov3Ds + 6 * idx;
  const REAL* mean   = means + 3 * idx;
  REAL dL_dconic[3]  = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
  REAL t[3];
  transformPoint4x3_colmap(mean, vm, t);
  const REAL limx = RC(1.3f) * tan_fovx;
  const REAL limy = RC(1.3f) * tan_fovy;
  const REAL txtz = t[0] / t[2];
  const REAL tytz = t[1] / t[2];
  t[0]            = r_min(limx, r_max(-limx, txtz)) * t[2];
  t[1]            = r_min(limy, r_max(-limy, tytz)) * t[2];
  const REAL x_grad_mul = (txtz < -limx || txtz > limx) ? RC(0) : RC(1);
  const REAL y_grad_mul = (tytz < -limy || tytz > limy) ? RC(0) : RC(1);
  gmat3 J   = gmat3_make(h_x / t[2], 0, -(h_x * t[0]) / (t[2] * t[2]), 0, h_y / t[2], -(h_y * t[1]) / (t[2] * t[2]), 0, 0, 0);
  gmat3 W   = gmat3_make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
  gmat3 Vrk = gmat3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  gmat3 T   = gmat3_mul(W, J);
  gmat3 cov2D = gmat3_mul(gmat3_mul(gmat3_transpose(T), gmat3_transpose(Vrk)), T);
  REAL a = cov2D.m[0][0] += RC(0.3f);
  REAL b = cov2D.m[0][1];
  REAL c = cov2D.m[1][1] += RC(0.3f);
  REAL denom = a * c - b * b;
  REAL dL_da = 0, dL_db = 0, dL_dc = 0;
  REAL denom2inv = RC(1.0) / ((denom * denom) + RC(0.0000001f));
#define TT(i, j) T.m[i][j]
#define VV(i, j) Vrk.m[i][j]
#define WW(i, j) W.m[i][j]
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
    dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
    dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
    dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
    dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
    dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
    dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db +
                           2 * TT(1, 0) * TT(1, 1) * dL_dc;
    dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db +
                           2 * TT(1, 0) * TT(1, 2) * dL_dc;
    dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db +
                           2 * TT(1, 1) * TT(1, 2) * dL_dc;
  } else {
    for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
  }
  REAL dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                 (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
  REAL dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                 (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
  REAL dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                 (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
  REAL dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                 (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
  REAL dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                 (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
  REAL dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                 (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
  REAL dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
  REAL dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
  REAL dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
  REAL dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
  REAL tz  = RC(1.) / t[2];
  REAL tz2 = tz * tz;
  REAL tz3 = tz2 * tz;
  REAL dL_dt[3];
  dL_dt[0] = x_grad_mul * -h_x * tz2 * dL_dJ02;
  dL_dt[1] = y_grad_mul * -h_y * tz2 * dL_dJ12;
  dL_dt[2] = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
  REAL dL_dmean[3];
  transformVec4x3Transpose_colmap(dL_dt, vm, dL_dmean);
  dL_dmeans[3 * idx + 0] = dL_dmean[0];
  dL_dmeans[3 * idx + 1] = dL_dmean[1];
  dL_dmeans[3 * idx + 2] = dL_dmean[2];
}

/* gaussian_preprocess.cu:183-298, restated literally INCLUDING its self-inconsistencies
 * (dL_dcov[3] uses T[1]; dL_dT2/5/8 = 0 so dL_dtx = dL_dty = 0). */
static void cov2D_backward_rowmajor(int idx, const REAL* means, const REAL* cov3Ds, REAL fx, REAL fy, REAL tan_fovx,
    REAL tan_fovy, const REAL* vm, const REAL* dL_dconics, REAL* dL_dmeans, REAL* dL_dcov) {
  const REAL* cov3D = cov3Ds + 6 * idx;
  const REAL* mean  = means + 3 * idx;
  REAL dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
  REAL t[3];
  xfm_p_4x3(mean, vm, t);
  const REAL limx = RC(1.3f) * tan_fovx;
  const REAL limy = RC(1.3f) * tan_fovy;
  const REAL txtz = t[0] / t[2];
  const REAL tytz = t[1] / t[2];
  t[0]            = r_min(r_max(txtz, -limx), limx) * t[2];
  t[1]            = r_min(r_max(tytz, -limy), limy) * t[2];
  const REAL x_grad_mul = (txtz < -limx || txtz > limx) ? RC(0) : RC(1);
  const REAL y_grad_mul = (tytz < -limy || tytz > limy) ? RC(0) : RC(1);
  REAL J[9]   = {fx / t[2], 0, -(fx * t[0]) / (t[2] * t[2]), 0, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0};
  REAL W[9]   = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  REAL Vrk[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
  REAL T[9]   = {0};
  matmul_3x3x3(W, J, T);
  REAL tmp[9] = {0};
  matmul_3x3x3_tn(T, Vrk, tmp);
  REAL cov2D[9] = {0};
  matmul_3x3x3(tmp, T, cov2D);
  REAL a = cov2D[0] += RC(0.3f);
  REAL b = cov2D[1];
  REAL c = cov2D[4] += RC(0.3f);
  REAL denom = a * c - b * b;
  REAL dL_da = 0, dL_db = 0, dL_dc = 0;
  REAL denom2inv = RC(1.0) / ((denom * denom) + RC(0.0000001f));
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
    dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
    dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
    dL_dcov[6 * idx + 0] = (T[0] * T[0] * dL_da + T[0] * T[1] * dL_db + T[1] * T[1] * dL_dc);
    dL_dcov[6 * idx + 3] = (T[1] * T[1] * dL_da + T[1] * T[4] * dL_db + T[4] * T[4] * dL_dc);
    dL_dcov[6 * idx + 5] = (T[6] * T[6] * dL_da + T[6] * T[7] * dL_db + T[7] * T[7] * dL_dc);
    dL_dcov[6 * idx + 1] = 2 * T[0] * T[3] * dL_da + (T[0] * T[4] + T[3] * T[1]) * dL_db + 2 * T[1] * T[4] * dL_dc;
    dL_dcov[6 * idx + 2] = 2 * T[0] * T[6] * dL_da + (T[0] * T[7] + T[6] * T[1]) * dL_db + 2 * T[1] * T[7] * dL_dc;
    dL_dcov[6 * idx + 4] = 2 * T[6] * T[3] * dL_da + (T[3] * T[7] + T[6] * T[4]) * dL_db + 2 * T[4] * T[7] * dL_dc;
  } else {
    for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
  }
  REAL dL_dT0 = 2 * (T[0] * cov3D[0] + T[3] * cov3D[1] + T[6] * cov3D[2]) * dL_da +
                (T[1] * cov3D[0] + T[4] * cov3D[1] + T[7] * cov3D[2]) * dL_db;
  REAL dL_dT1 = 2 * (T[1] * cov3D[0] + T[4] * cov3D[1] + T[7] * cov3D[2]) * dL_dc +
                (T[0] * cov3D[0] + T[3] * cov3D[1] + T[6] * cov3D[2]) * dL_db;
  REAL dL_dT2 = 0;
  REAL dL_dT3 = 2 * (T[0] * cov3D[1] + T[3] * cov3D[3] + T[6] * cov3D[4]) * dL_da +
                (T[1] * cov3D[1] + T[4] * cov3D[3] + T[7] * cov3D[4]) * dL_db;
  REAL dL_dT4 = 2 * (T[1] * cov3D[1] + T[4] * cov3D[3] + T[7] * cov3D[4]) * dL_dc +
                (T[0] * cov3D[3] + T[3] * cov3D[4] + T[6] * cov3D[5]) * dL_db;
  REAL dL_dT5 = 0;
  REAL dL_dT6 = 2 * (T[0] * cov3D[2] + T[3] * cov3D[4] + T[6] * cov3D[5]) * dL_da +
                (T[1] * cov3D[2] + T[4] * cov3D[4] + T[7] * cov3D[5]) * dL_db;
  REAL dL_dT7 = 2 * (T[1] * cov3D[2] + T[4] * cov3D[4] + T[7] * cov3D[5]) * dL_dc +
                (T[0] * cov3D[2] + T[3] * cov3D[4] + T[6] * cov3D[5]) * dL_db;
  REAL dL_dT8 = 0;
  REAL dL_dJ00 = W[0] * dL_dT0 + W[3] * dL_dT3 + W[6] * dL_dT6;
  REAL dL_dJ02 = W[0] * dL_dT2 + W[3] * dL_dT5 + W[6] * dL_dT8;
  REAL dL_dJ11 = W[1] * dL_dT1 + W[4] * dL_dT4 + W[7] * dL_dT7;
  REAL dL_dJ12 = W[1] * dL_dT2 + W[4] * dL_dT5 + W[7] * dL_dT8;
  REAL tz  = RC(1.) / t[2];
  REAL tz2 = tz * tz;
  REAL tz3 = tz2 * tz;
  REAL dL_dt[3];
  dL_dt[0] = x_grad_mul * -fx * tz2 * dL_dJ02;
  dL_dt[1] = y_grad_mul * -fy * tz2 * dL_dJ12;
  dL_dt[2] = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t[0]) * tz3 * dL_dJ02 + (2 * fy * t[1]) * tz3 * dL_dJ12;
  REAL dL_dmean[3];
  xfm_v_4x3_T(dL_dt, vm, dL_dmean);
  dL_dmeans[3 * idx + 0] = dL_dmean[0];
  dL_dmeans[3 * idx + 1] = dL_dmean[1];
  dL_dmeans[3 * idx + 2] = dL_dmean[2];
}

/* computeCov3D_colmap bwd: gaussian_preprocess_colmap.cu:357-420 */
static void cov3D_backward_colmap(int idx, const REAL* scale, REAL mod, const REAL* rot, const REAL* dL_dcov3Ds,
    REAL* dL_dscales, REAL* dL_drots) {
  REAL x = rot[0], y = rot[1], z = rot[2], r = rot[3];
  gmat3 R = colmap_R(rot);
  gmat3 S = gmat3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
  REAL s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.m[0][0] = s[0], S.m[1][1] = s[1], S.m[2][2] = s[2];
  gmat3 M             = gmat3_mul(S, R);
  const REAL* dL_dcov = dL_dcov3Ds + 6 * idx;
  gmat3 dL_dSigma = gmat3_make(dL_dcov[0], RC(0.5) * dL_dcov[1], RC(0.5) * dL_dcov[2], RC(0.5) * dL_dcov[1], dL_dcov[3],
      RC(0.5) * dL_dcov[4], RC(0.5) * dL_dcov[2], RC(0.5) * dL_dcov[4], dL_dcov[5]);
  /* dL_dM = 2.0f * M * dL_dSigma : (scalar * mat) first, glm evaluates left to right */
  gmat3 M2;
  for (int c = 0; c < 3; ++c)
    for (int w = 0; w < 3; ++w) M2.m[c][w] = M.m[c][w] * RC(2.0);
  gmat3 dL_dM  = gmat3_mul(M2, dL_dSigma);
  gmat3 Rt     = gmat3_transpose(R);
  gmat3 dL_dMt = gmat3_transpose(dL_dM);
  for (int k = 0; k < 3; ++k)
    dL_dscales[3 * idx + k] = Rt.m[k][0] * dL_dMt.m[k][0] + Rt.m[k][1] * dL_dMt.m[k][1] + Rt.m[k][2] * dL_dMt.m[k][2];
  for (int k = 0; k < 3; ++k)
    for (int w = 0; w < 3; ++w) dL_dMt.m[k][w] *= s[k];
#define D(i, j) dL_dMt.m[i][j]
  REAL dq[4];
  dq[0] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
  dq[1] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
  dq[2] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
  dq[3] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
#undef D
  for (int k = 0; k < 4; ++k) dL_drots[4 * idx + k] = dq[k];
}

/* computeCov3D_backward: gaussian_preprocess.cu:301-336 */
static void cov3D_backward_rowmajor(int idx, const REAL* scale, const REAL* rot, const REAL* dL_dcov3Ds,
    REAL* dL_dscales, REAL* dL_drots) {
  REAL R[9] = {0};
  quaternion_to_R(rot, R);
  const REAL* g = dL_dcov3Ds + 6 * idx;
  REAL gs[3];
  gs[0] = R[0] * R[0] * g[0] + R[0] * R[3] * g[1] + R[0] * R[6] * g[2] + R[3] * R[3] * g[3] + R[3] * R[6] * g[4] + R[6] * R[6] * g[5];
  gs[1] = R[1] * R[1] * g[0] + R[1] * R[4] * g[1] + R[1] * R[7] * g[2] + R[4] * R[4] * g[3] + R[4] * R[7] * g[4] + R[7] * R[7] * g[5];
  gs[2] = R[2] * R[2] * g[0] + R[2] * R[5] * g[1] + R[2] * R[8] * g[2] + R[5] * R[5] * g[3] + R[5] * R[8] * g[4] + R[8] * R[8] * g[5];
  gs[0] *= 2 * scale[0];
  gs[1] *= 2 * scale[1];
  gs[2] *= 2 * scale[2];
  dL_dscales[3 * idx] = gs[0], dL_dscales[3 * idx + 1] = gs[1], dL_dscales[3 * idx + 2] = gs[2];
  REAL sx2 = scale[0] * scale[0], sy2 = scale[1] * scale[1], sz2 = scale[2] * scale[2];
  REAL dR[9];
  dR[0] = (2 * R[0] * g[0] + R[3] * g[1] + R[6] * g[2]) * sx2;
  dR[1] = (2 * R[1] * g[0] + R[4] * g[1] + R[7] * g[2]) * sy2;
  dR[2] = (2 * R[2] * g[0] + R[5] * g[1] + R[8] * g[2]) * sz2;
  dR[3] = (2 * R[3] * g[3] + R[0] * g[1] + R[6] * g[4]) * sx2;
  dR[4] = (2 * R[4] * g[3] + R[1] * g[1] + R[7] * g[4]) * sy2;
  dR[5] = (2 * R[5] * g[3] + R[2] * g[1] + R[8] * g[4]) * sz2;
  dR[6] = (2 * R[6] * g[5] + R[0] * g[2] + R[3] * g[4]) * sx2;
  dR[7] = (2 * R[7] * g[5] + R[1] * g[2] + R[4] * g[4]) * sy2;
  dR[8] = (2 * R[8] * g[5] + R[2] * g[2] + R[5] * g[4]) * sz2;
  dL_quaternion_to_R(rot, dR, dL_drots + 4 * idx);
}

/* preprocess_backward{,_colmap}: computeCov2DCUDA then preprocessCUDA_backward.
 * gaussian_preprocess.cu:340-400, gaussian_preprocess_colmap.cu:424-481.
 * All gradient outputs must be zero-initialised by the caller (torch::zeros in the reference);
 * Gaussians with radii<=0 are skipped entirely. cov3Ds = cov3D_precomp if given else the forward's cov3D. */
void ORACLE(preprocess_backward)(int P, int D, int M, const REAL* means3D, const int32_t* radii, const REAL* shs,
    const uint8_t* clamped, const REAL* scales, const REAL* rotations, REAL scale_modifier, const REAL* cov3Ds,
    const REAL* viewmatrix, const REAL* projmatrix, int W, int H, REAL tan_fovx, REAL tan_fovy, const REAL* campos,
    int colmap, const REAL* dL_dmean2D /*[P,3]*/, const REAL* dL_dconic /*[P,4]*/, REAL* dL_dmean3D, REAL* dL_dcolor,
    REAL* dL_dcov3D, REAL* dL_dsh, REAL* dL_dscale, REAL* dL_drot) {
  const REAL focal_y = H / (RC(2.0) * tan_fovy);
  const REAL focal_x = W / (RC(2.0) * tan_fovx);
  const REAL* proj   = projmatrix;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    if (!(radii[idx] > 0)) continue;
    if (colmap)
      cov2D_backward_colmap(idx, means3D, cov3Ds, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic, dL_dmean3D, dL_dcov3D);
    else
      cov2D_backward_rowmajor(idx, means3D, cov3Ds, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic, dL_dmean3D, dL_dcov3D);
    const REAL* m = means3D + 3 * idx;
    REAL m_hom[4];
    REAL dL_dmean[3];
    const REAL gx2 = dL_dmean2D[3 * idx], gy2 = dL_dmean2D[3 * idx + 1];
    if (colmap) {
      transformPoint4x4_colmap(m, proj, m_hom);
      REAL m_w  = RC(1.0) / (m_hom[3] + RC(0.0000001f));
      REAL mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
      REAL mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
      dL_dmean[0] = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
      dL_dmean[1] = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
      dL_dmean[2] = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    } else {
      xfm_p_4x4(m, proj, m_hom);
      REAL m_w  = RC(1.0) / (m_hom[3] + RC(0.0000001f));
      REAL mul1 = (proj[0] * m[0] + proj[1] * m[1] + proj[2] * m[2] + proj[3]) * m_w * m_w;
      REAL mul2 = (proj[4] * m[0] + proj[5] * m[1] + proj[6] * m[2] + proj[7]) * m_w * m_w;
      dL_dmean[0] = (proj[0] * m_w - proj[12] * mul1) * gx2 + (proj[4] * m_w - proj[12] * mul2) * gy2;
      dL_dmean[1] = (proj[1] * m_w - proj[13] * mul1) * gx2 + (proj[5] * m_w - proj[13] * mul2) * gy2;
      dL_dmean[2] = (proj[2] * m_w - proj[14] * mul1) * gx2 + (proj[6] * m_w - proj[14] * mul2) * gy2;
    }
    dL_dmean3D[3 * idx + 0] += dL_dmean[0];
    dL_dmean3D[3 * idx + 1] += dL_dmean[1];
    dL_dmean3D[3 * idx + 2] += dL_dmean[2];
    if (shs) computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmean3D, dL_dsh);
    if (scales) {
      if (colmap)
        cov3D_backward_colmap(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
      else
        cov3D_backward_rowmajor(idx, scales + 3 * idx, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
    }
  }
}

/* ================================================================================================
 * render_extra_forward_kernel: gaussian_rasterizer_extra.cu:10-98.  pixel_extra is [H*W, E] (pix-major).
 * ============================================================================================== */
void ORACLE(render_extra_forward)(int W, int H, int E, const uint32_t* ranges, const uint32_t* point_list,
    const REAL* means2D, const REAL* conic_opacity, const uint32_t* n_contrib, const REAL* point_extra,
    REAL* pixel_extra) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        const int last_contributor = (int) n_contrib[pix_id];
        REAL* out = pixel_extra + (size_t) pix_id * E;
        for (int e = 0; e < E; ++e) out[e] = 0;
        REAL T               = RC(1.0);
        uint32_t contributor = 0;
        for (uint32_t k = r0; k < r1; ++k) {
          contributor++;
          if ((int64_t) contributor > (int64_t) last_contributor) break;
          const uint32_t id = point_list[k];
          REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const REAL* con_o = conic_opacity + 4 * id;
          REAL power        = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
          if (power > RC(0.0)) continue;
          REAL alpha = r_min(RC(0.99f), con_o[3] * oexp(power));
          if (alpha < RC(1.0f / 255.0f)) continue;
          REAL test_T = T * (1 - alpha);
          if (test_T < RC(0.0001f)) break;
          for (int e = 0; e < E; ++e) out[e] += point_extra[(size_t) id * E + e] * alpha * T;
          T = test_T;
        }
      }
  }
}

/* render_extra_backward_kernel: gaussian_rasterizer_extra.cu:100-220.  No dL_dT term; accumulates INTO
 * dL_dmean2D [P,3], dL_dconic2D [P,4], dL_dopacity [P]; dL_dpoint_extra [P,E] zero-initialised by caller. */
void ORACLE(render_extra_backward)(int P, int W, int H, int E, const uint32_t* ranges, const uint32_t* point_list,
    const REAL* means2D, const REAL* conic_opacity, const REAL* out_opacity, const uint32_t* n_contrib,
    const REAL* point_extra, const REAL* dL_dpixel_extra, REAL* dL_dmean2D, REAL* dL_dconic2D, REAL* dL_dopacity,
    REAL* dL_dpoint_extra) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const REAL ddelx_dx = (REAL) (0.5 * W);
  const REAL ddely_dy = (REAL) (0.5 * H);
  (void) P;
  /* sequential over tiles: this path is small in tests; keeps accumulation order fixed */
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        const REAL T_final    = RC(1.0) - out_opacity[pix_id];
        const int last_contributor = (int) n_contrib[pix_id];
        for (int es = 0; es < E; es += 16) {
          uint32_t contributor = r1 - r0;
          REAL T               = T_final;
          REAL accum_rec[16] = {0}, last_extra[16] = {0}, dL_dpixel[16];
          for (int e = 0; e < 16; ++e) dL_dpixel[e] = (e + es < E) ? dL_dpixel_extra[(size_t) pix_id * E + e + es] : 0;
          REAL last_alpha = 0;
          for (uint32_t k = r1; k-- > r0;) {
            contributor--;
            if ((int64_t) contributor >= (int64_t) last_contributor) continue;
            const uint32_t id = point_list[k];
            const REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
            const REAL* con_o = conic_opacity + 4 * id;
            const REAL power  = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
            if (power > RC(0.0)) continue;
            const REAL G     = oexp(power);
            const REAL alpha = r_min(RC(0.99f), con_o[3] * G);
            if (alpha < RC(1.0f / 255.0f)) continue;
            T                          = T / (RC(1.) - alpha);
            const REAL dchannel_dextra = alpha * T;
            REAL dL_dalpha             = RC(0.0);
            for (int e = 0; e < 16; ++e) {
              if (e + es >= E) continue;
              const REAL c  = point_extra[(size_t) id * E + e + es];
              accum_rec[e]  = last_alpha * last_extra[e] + (RC(1.) - last_alpha) * accum_rec[e];
              last_extra[e] = c;
              dL_dalpha += (c - accum_rec[e]) * dL_dpixel[e];
              dL_dpoint_extra[(size_t) id * E + e + es] += dchannel_dextra * dL_dpixel[e];
            }
            dL_dalpha *= T;
            last_alpha = alpha;
            const REAL dL_dG    = con_o[3] * dL_dalpha;
            const REAL gdx      = G * dx;
            const REAL gdy      = G * dy;
            const REAL dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
            const REAL dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
            dL_dmean2D[3 * id + 0] += dL_dG * dG_ddelx * ddelx_dx;
            dL_dmean2D[3 * id + 1] += dL_dG * dG_ddely * ddely_dy;
            dL_dconic2D[4 * id + 0] += RC(-0.5) * gdx * dx * dL_dG;
            dL_dconic2D[4 * id + 1] += RC(-0.5) * gdx * dy * dL_dG;
            dL_dconic2D[4 * id + 3] += RC(-0.5) * gdy * dy * dL_dG;
            dL_dopacity[id] += G * dL_dalpha;
          }
        }
      }
  }
}

/* render_topk_weights: gaussian_topk.cu:10-96.  top_indices [H,W,k] prefilled -1, top_weights prefilled 0.
 * NOTE the reference skips entries with contributor >= last_contributor (so the last contributor itself is
 * never ranked); restated literally. */
void ORACLE(topk_weights)(int topk, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
    const REAL* means2D, const REAL* conic_opacity, const uint32_t* n_contrib, int32_t* top_indices,
    REAL* top_weights) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; ++tile) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ++ly)
      for (int lx = 0; lx < BLOCK_X; ++lx) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t) W * py + px;
        const REAL pixf[2]    = {(REAL) px, (REAL) py};
        REAL* tw    = top_weights + (size_t) pix_id * topk;
        int32_t* ti = top_indices + (size_t) pix_id * topk;
        REAL T = RC(1.0);
        uint32_t contributor = 0;
        const int last_contributor = (int) n_contrib[pix_id];
        for (uint32_t k = r0; k < r1; ++k) {
          contributor++;
          if ((int64_t) contributor >= (int64_t) last_contributor) continue;
          const uint32_t id = point_list[k];
          REAL dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const REAL* con_o = conic_opacity + 4 * id;
          REAL power        = RC(-0.5) * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
          if (power > RC(0.0)) continue;
          REAL alpha = r_min(RC(0.99f), con_o[3] * oexp(power));
          if (alpha < RC(1.0f / 255.0f)) continue;
          REAL test_T = T * (1 - alpha);
          if (test_T < RC(0.0001f)) break;
          REAL w      = alpha * T;
          int32_t idx = (int32_t) id;
          for (int q = 0; q < topk; ++q) {
            if (w >= tw[q]) {
              REAL t0   = tw[q];
              tw[q]     = w;
              w         = t0;
              int32_t i0 = ti[q];
              ti[q]      = idx;
              idx        = i0;
            }
          }
          T = test_T;
        }
      }
  }
}

/* ================================================================================================
 * LBS deform + activation epilogue.
 *   networks/sk_gs.py:1147-1149 (sk_stage) / :813-825 (warp):
 *       d_xyz   = sum_k w[n,k] * act(T[idx[n,k]], p_n) - p_n
 *       d_rot   = sum_k w[n,k] * bone_drot[idx[n,k]]
 *       d_scale = sum_k w[n,k] * bone_dscale[idx[n,k]]
 *   networks/sk_gs.py:1162,1192,1202-1203 (forward):
 *       opacity = sigmoid(opacity_logit); means = xyz + d_xyz; scales = exp(log_scale) + d_scale;
 *       rotations = F.normalize(rot + d_rot)   (eps 1e-12)
 *   SE3 act: C/include/lie.h:45-47 (ctor normalises q), :59-64 (p + w*uv + q x uv, uv = 2 q x p), :246 (+ t).
 *   bone_T is [M,7] = [tx,ty,tz,qx,qy,qz,qw].  `points` is the DETACHED copy of xyz (sk_gs.py:1113).
 * ============================================================================================== */
static void se3_act(const REAL* T7, const REAL* p, REAL* out) {
  const REAL* t = T7;
  REAL q[4]     = {T7[3], T7[4], T7[5], T7[6]};
  REAL n        = R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
  REAL uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
  REAL c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  out[0]    = p[0] + q[3] * uv[0] + c[0] + t[0];
  out[1]    = p[1] + q[3] * uv[1] + c[1] + t[1];
  out[2]    = p[2] + q[3] * uv[2] + c[2] + t[2];
}

void ORACLE(lbs_deform_forward)(int P, int K, int M, const REAL* points, const REAL* weights, const int64_t* indices,
    const REAL* bone_T, const REAL* bone_drot, const REAL* bone_dscale, const REAL* xyz, const REAL* log_scale,
    const REAL* rot, const REAL* opacity_logit, REAL* means, REAL* scales, REAL* rotations, REAL* opacity, REAL* d_xyz,
    REAL* d_rot, REAL* d_scale) {
  (void) M;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < P; ++n) {
    const REAL* p = points + 3 * n;
    REAL sx[3] = {0, 0, 0}, sr[4] = {0, 0, 0, 0}, ss[3] = {0, 0, 0};
    for (int k = 0; k < K; ++k) {
      const int64_t j = indices[(size_t) n * K + k];
      const REAL w    = weights[(size_t) n * K + k];
      REAL y[3];
      se3_act(bone_T + 7 * j, p, y);
      for (int c = 0; c < 3; ++c) sx[c] += y[c] * w;
      for (int c = 0; c < 4; ++c) sr[c] += bone_drot[4 * j + c] * w;
      for (int c = 0; c < 3; ++c) ss[c] += bone_dscale[3 * j + c] * w;
    }
    REAL v[4];
    for (int c = 0; c < 3; ++c) {
      REAL dx = sx[c] - p[c];
      if (d_xyz) d_xyz[3 * n + c] = dx;
      means[3 * n + c]  = xyz[3 * n + c] + dx;
      scales[3 * n + c] = R_EXP(log_scale[3 * n + c]) + ss[c];
      if (d_scale) d_scale[3 * n + c] = ss[c];
    }
    for (int c = 0; c < 4; ++c) {
      v[c] = rot[4 * n + c] + sr[c];
      if (d_rot) d_rot[4 * n + c] = sr[c];
    }
    REAL nv = R_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    nv      = r_max(nv, RC(1e-12));
    for (int c = 0; c < 4; ++c) rotations[4 * n + c] = v[c] / nv;
    opacity[n] = RC(1.0) / (RC(1.0) + R_EXP(-opacity_logit[n]));
  }
}

/* Backward of the above.  Euclidean gradients w.r.t. every input; for bone_T the gradient includes the
 * Jacobian of the internal quaternion normalisation, which makes it orthogonal to q -- i.e. exactly the
 * embedding gradient lietorch's FromVec projector produces (C/include/lie.h:82-90,303-311; derivation in
 * DESIGN.md).  g_bone_* are accumulated sequentially over n (deterministic); caller zero-initialises them. */
void ORACLE(lbs_deform_backward)(int P, int K, int M, const REAL* points, const REAL* weights, const int64_t* indices,
    const REAL* bone_T, const REAL* bone_drot, const REAL* bone_dscale, const REAL* log_scale, const REAL* rot,
    const REAL* opacity_logit, const REAL* g_means, const REAL* g_scales, const REAL* g_rotations,
    const REAL* g_opacity, REAL* g_weights, REAL* g_bone_T, REAL* g_bone_drot, REAL* g_bone_dscale, REAL* g_xyz,
    REAL* g_log_scale, REAL* g_rot, REAL* g_opacity_logit) {
  (void) M;
  for (int n = 0; n < P; ++n) {
    const REAL* p = points + 3 * n;
    /* recompute d_rot for the normalisation backward */
    REAL sr[4] = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
      const int64_t j = indices[(size_t) n * K + k];
      const REAL w    = weights[(size_t) n * K + k];
      for (int c = 0; c < 4; ++c) sr[c] += bone_drot[4 * j + c] * w;
    }
    REAL v[4];
    for (int c = 0; c < 4; ++c) v[c] = rot[4 * n + c] + sr[c];
    REAL nv  = R_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    REAL g_v[4];
    if (nv > RC(1e-12)) {
      REAL u[4] = {v[0] / nv, v[1] / nv, v[2] / nv, v[3] / nv};
      REAL dot  = u[0] * g_rotations[4 * n] + u[1] * g_rotations[4 * n + 1] + u[2] * g_rotations[4 * n + 2] +
                 u[3] * g_rotations[4 * n + 3];
      for (int c = 0; c < 4; ++c) g_v[c] = (g_rotations[4 * n + c] - u[c] * dot) / nv;
    } else {
      for (int c = 0; c < 4; ++c) g_v[c] = g_rotations[4 * n + c] / RC(1e-12);
    }
    const REAL* g_dx = g_means + 3 * n;  /* d means / d d_xyz = I */
    const REAL* g_ds = g_scales + 3 * n; /* d scales / d d_scale = I */
    for (int c = 0; c < 3; ++c) {
      g_xyz[3 * n + c]       = g_means[3 * n + c];
      g_log_scale[3 * n + c] = g_scales[3 * n + c] * R_EXP(log_scale[3 * n + c]);
    }
    for (int c = 0; c < 4; ++c) g_rot[4 * n + c] = g_v[c];
    REAL sg            = RC(1.0) / (RC(1.0) + R_EXP(-opacity_logit[n]));
    g_opacity_logit[n] = g_opacity[n] * sg * (RC(1.0) - sg);
    for (int k = 0; k < K; ++k) {
      const int64_t j = indices[(size_t) n * K + k];
      const REAL w    = weights[(size_t) n * K + k];
      const REAL* T7  = bone_T + 7 * j;
      REAL y[3];
      se3_act(T7, p, y);
      REAL gw = g_dx[0] * y[0] + g_dx[1] * y[1] + g_dx[2] * y[2];
      for (int c = 0; c < 4; ++c) gw += g_v[c] * bone_drot[4 * j + c];
      for (int c = 0; c < 3; ++c) gw += g_ds[c] * bone_dscale[3 * j + c];
      g_weights[(size_t) n * K + k] = gw;
      for (int c = 0; c < 4; ++c) g_bone_drot[4 * j + c] += w * g_v[c];
      for (int c = 0; c < 3; ++c) g_bone_dscale[3 * j + c] += w * g_ds[c];
      /* g wrt y */
      REAL g[3] = {w * g_dx[0], w * g_dx[1], w * g_dx[2]};
      for (int c = 0; c < 3; ++c) g_bone_T[7 * j + c] += g[c];
      /* y = p + 2 qw (qv x p) + 2 qv x (qv x p) evaluated at the unit quaternion qh = q/|q| */
      REAL q[4] = {T7[3], T7[4], T7[5], T7[6]};
      REAL qn   = R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      REAL qh[4] = {q[0] / qn, q[1] / qn, q[2] / qn, q[3] / qn};
      const REAL* vq = qh;
      REAL vxp[3] = {vq[1] * p[2] - vq[2] * p[1], vq[2] * p[0] - vq[0] * p[2], vq[0] * p[1] - vq[1] * p[0]};
      REAL pxg[3] = {p[1] * g[2] - p[2] * g[1], p[2] * g[0] - p[0] * g[2], p[0] * g[1] - p[1] * g[0]};
      REAL vdp = vq[0] * p[0] + vq[1] * p[1] + vq[2] * p[2];
      REAL gdv = g[0] * vq[0] + g[1] * vq[1] + g[2] * vq[2];
      REAL gdp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
      REAL gqh[4];
      for (int c = 0; c < 3; ++c)
        gqh[c] = RC(2.0) * qh[3] * pxg[c] + RC(2.0) * (vdp * g[c] + gdv * p[c] - RC(2.0) * gdp * vq[c]);
      gqh[3] = RC(2.0) * (g[0] * vxp[0] + g[1] * vxp[1] + g[2] * vxp[2]);
      REAL dotq = qh[0] * gqh[0] + qh[1] * gqh[1] + qh[2] * gqh[2] + qh[3] * gqh[3];
      for (int c = 0; c < 4; ++c) g_bone_T[7 * j + 3 + c] += (gqh[c] - qh[c] * dotq) / qn;
    }
  }
}

/* K nearest bones, squared L2, ascending (pytorch3d.ops.knn_points semantics as used at sk_gs.py:757);
 * ties resolved towards the lower bone index.  dim = 3 (or 3+F when hyper features are concatenated). */
void ORACLE(knn_bones)(int P, int M, int K, int dim, const REAL* points, const REAL* joints, REAL* out_dist,
    int64_t* out_idx) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < P; ++n) {
    REAL bd[16];
    int64_t bi[16];
    int cnt = 0;
    for (int j = 0; j < M; ++j) {
      REAL d = 0;
      for (int c = 0; c < dim; ++c) {
        REAL df = points[(size_t) n * dim + c] - joints[(size_t) j * dim + c];
        d += df * df;
      }
      int pos = cnt;
      while (pos > 0 && bd[pos - 1] > d) --pos; /* strict: equal distances keep the lower index first */
      if (pos < K) {
        int last = cnt < K ? cnt : K - 1;
        for (int q = last; q > pos; --q) bd[q] = bd[q - 1], bi[q] = bi[q - 1];
        bd[pos] = d, bi[pos] = j;
        if (cnt < K) cnt++;
      }
    }
    for (int k = 0; k < K; ++k) {
      out_dist[(size_t) n * K + k] = k < cnt ? bd[k] : RC(0);
      out_idx[(size_t) n * K + k]  = k < cnt ? bi[k] : -1;
    }
  }
}

/* mark_visible: commented out in the reference (gaussian_rasterizer_imp.cu:75-103); semantics = the near-plane
 * test of in_frustum{,_colmap}. */
void ORACLE(mark_visible)(int P, const REAL* means3D, const REAL* viewmatrix, int colmap, uint8_t* present) {
  for (int idx = 0; idx < P; ++idx) {
    REAL pv[3];
    if (colmap) {
      transformPoint4x3_colmap(means3D + 3 * idx, viewmatrix, pv);
      present[idx] = !(pv[2] <= RC(0.2f));
    } else {
      xfm_p_4x3(means3D + 3 * idx, viewmatrix, pv);
      present[idx] = !(pv[2] <= RC(-1.0f));
    }
  }
}

/* ---- test hooks that expose sub-steps so they can be pinned against the reference's pure-torch helpers -------- */
/* Sigma3D [P,6] from scale/rotation, both modes (computeCov3D{,_colmap}) */
void ORACLE(cov3d_array)(int P, const REAL* scales, REAL mod, const REAL* rots, int colmap, REAL* out) {
  for (int i = 0; i < P; ++i) {
    if (colmap)
      computeCov3D_colmap(scales + 3 * i, mod, rots + 4 * i, out + 6 * i);
    else
      computeCov3D_rowmajor(scales + 3 * i, rots + 4 * i, out + 6 * i);
  }
}
/* Sigma2D (+0.3 low-pass) [P,3] = (xx, xy, yy), both modes (computeCov2D{,_colmap}) */
void ORACLE(cov2d_array)(int P, const REAL* means, const REAL* cov3D, const REAL* viewmatrix, REAL fx, REAL fy,
    REAL tan_fovx, REAL tan_fovy, int colmap, REAL* out) {
  for (int i = 0; i < P; ++i) {
    if (colmap)
      computeCov2D_colmap(means + 3 * i, fx, fy, tan_fovx, tan_fovy, cov3D + 6 * i, viewmatrix, out + 3 * i);
    else
      computeCov2D_rowmajor(means + 3 * i, fx, fy, tan_fovx, tan_fovy, cov3D + 6 * i, viewmatrix, out + 3 * i);
  }
}
/* SH -> RGB [P,3] (with the +0.5 and the clamp) and the clamp flags */
void ORACLE(sh_array)(int P, int deg, int M, const REAL* means, const REAL* campos, const REAL* shs, REAL* rgb,
    uint8_t* clamped) {
  for (int i = 0; i < P; ++i) computeColorFromSH_fwd(i, deg, M, means, campos, shs, clamped, rgb + 3 * i);
}

/* Bone chain, literal restatement of kinematic() + skeleton_warp_SE3() (networks/sk_gs.py:1069-1107,193-206) with the
 * SE3 product of lie.h:45-47,242-246: pointer jumping over the [M,L] ancestor table, root forced to identity.
 * sk_r [M,4] are the (already normalised) joint rotations; global_T may be NULL. */
static void se3_mul_o(const REAL* a, const REAL* b, REAL* o) {
  REAL qa[4] = {a[3], a[4], a[5], a[6]}, qb[4] = {b[3], b[4], b[5], b[6]};
  REAL na = R_SQRT(qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3]);
  REAL nb = R_SQRT(qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2] + qb[3] * qb[3]);
  for (int c = 0; c < 4; ++c) qa[c] /= na, qb[c] /= nb;
  /* Eigen quaternion product (x,y,z,w) */
  REAL q[4] = {qa[3] * qb[0] + qa[0] * qb[3] + qa[1] * qb[2] - qa[2] * qb[1],
      qa[3] * qb[1] + qa[1] * qb[3] + qa[2] * qb[0] - qa[0] * qb[2],
      qa[3] * qb[2] + qa[2] * qb[3] + qa[0] * qb[1] - qa[1] * qb[0],
      qa[3] * qb[3] - qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2]};
  REAL nq = R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  REAL ta[7] = {0, 0, 0, qa[0], qa[1], qa[2], qa[3]}, y[3];
  se3_act(ta, b, y); /* R(qa) * t_b */
  o[0] = a[0] + y[0], o[1] = a[1] + y[1], o[2] = a[2] + y[2];
  for (int c = 0; c < 4; ++c) o[3 + c] = q[c] / nq;
}
void ORACLE(bone_chain_forward)(int M, int L, int root, const int64_t* ancestors /*[M,L]*/, const REAL* sk_r,
    const REAL* joints, const REAL* global_T, REAL* bone_T) {
  REAL* cur = (REAL*) malloc(sizeof(REAL) * 7 * M);
  REAL* nxt = (REAL*) malloc(sizeof(REAL) * 7 * M);
  for (int i = 0; i < M; ++i) {
    REAL rot7[7] = {0, 0, 0, sk_r[4 * i], sk_r[4 * i + 1], sk_r[4 * i + 2], sk_r[4 * i + 3]};
    REAL mj[3]   = {-joints[3 * i], -joints[3 * i + 1], -joints[3 * i + 2]}, y[3];
    se3_act(rot7, mj, y); /* sk_t = joints + sk_r.act(-joints), sk_gs.py:1090 */
    for (int c = 0; c < 3; ++c) cur[7 * i + c] = joints[3 * i + c] + y[c];
    for (int c = 0; c < 4; ++c) cur[7 * i + 3 + c] = sk_r[4 * i + c];
  }
  for (int c = 0; c < 7; ++c) cur[7 * root + c] = (c == 6) ? RC(1) : RC(0); /* out[root] = identity, :196 */
  for (int l = 0; l < L; ++l) {                                             /* out = out[parents[:, l]] * out, :199 */
    for (int i = 0; i < M; ++i) se3_mul_o(cur + 7 * ancestors[(size_t) i * L + l], cur + 7 * i, nxt + 7 * i);
    REAL* t = cur;
    cur     = nxt;
    nxt     = t;
  }
  for (int i = 0; i < M; ++i) {
    if (global_T)
      se3_mul_o(global_T, cur + 7 * i, bone_T + 7 * i);
    else
      for (int c = 0; c < 7; ++c) bone_T[7 * i + c] = cur[7 * i + c];
  }
  free(cur);
  free(nxt);
}

/* test hook: out[i] = the blend exp (current exp_mode) of x[i] */
void ORACLE(exp_array)(int n, const REAL* x, REAL* out) {
  for (int i = 0; i < n; ++i) out[i] = oexp(x[i]);
}

/* test / bench hook: number of OpenMP threads of the following calls (the single-thread CPU baseline) */
void ORACLE(set_num_threads)(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void) n;
#endif
}

int ORACLE(num_threads)(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
