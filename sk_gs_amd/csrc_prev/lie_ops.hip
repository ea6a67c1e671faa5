// lie_ops.hip -- the Lie-group operators of the reference's deform (lietorch's backend: exp, log, inv, mul, adj, adjT, act, act4
// and the embedding <-> tangent maps at InitFromVec / vec()), forward and backward, SO3 and SE3: one launch per operator call,
// one lane per group element.  This is what `lietorch_backends` is to upstream lietorch and what
// my_ext/_C/src/ops_3d/lie_{cpu.cpp,gpu.cu,torch.cpp} (pybind `lie_expm ... lie_act4`, lie_torch.cpp:357-385) restate in the
// reference tree; sk_gs_amd/lietorch.py calls it for tensors on a HIP device (its pure-torch bodies, the same formulas, remain the
// CPU path: ~40-60 torch kernels per operator, which made the 20-bone kinematic chain of stage `sk` 30 ms of launches per step).
//
// Semantics (file:line of the reference's copy of lietorch's C++ core):
//   forward   lie.h:45-64 (SO3 ctor NORMALISES q; q*q; act = p + w uv + q x uv), :107-176 (Log, Exp, left Jacobian and inverse),
//             :232-252 (SE3 ctor, inv, product, act, act4), :254-263 (Adj), :314-385 (SE3 Log / Exp / Q / Jacobians)
//   backward  lie_cpu.cpp:25-38 exp: da = dX J_l(a);   :54-67 log: dX = da J_l^-1(log X);   :84-97 inv: dX = -dY Adj(X^-1);
//             :111-126 mul: dX = dZ, dY = dZ Adj(X);   :143-162 adj: da = db Adj(X), dX = -db adj(Adj a);
//             :179-198 adjT: da = Adj(X) db, dX = -a adj(Adj(X) db);   :217-236 act: dp = dq R, dX = dq [I | -hat(X p)];
//             :288-309 act4.  Gradients of group elements are LEFT-TANGENT row vectors in the first K of N slots, the rest zero.
//   ToVec / FromVec backward (upstream's Python glue over orthogonal_projector, lie.h:82-90,303-311): g J and g pinv(J), the
//             latter in closed form (J_q^T J_q = I/4 for a unit quaternion): (tau, 4 J_q (phi - t x tau)).
#include "skgs_common.h"

#pragma clang fp contract(off)

namespace skgs {
namespace lie {

constexpr float EPS = 1e-6f;  // lie.h:23
constexpr float PI_F = 3.14159265358979323846f;

struct V3 {
  float x, y, z;
};
struct M3 {
  float m[3][3];
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ M3 hat(V3 v) {
  M3 r = {{{0.f, -v.z, v.y}, {v.z, 0.f, -v.x}, {-v.y, v.x, 0.f}}};
  return r;
}
__device__ __forceinline__ M3 eye() {
  M3 r = {{{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}};
  return r;
}
__device__ __forceinline__ M3 mm(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
__device__ __forceinline__ M3 add(const M3& a, const M3& b, float sb = 1.f) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + sb * b.m[i][j];
  return r;
}
__device__ __forceinline__ M3 scale(float s, const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j];
  return r;
}
__device__ __forceinline__ V3 mv(const M3& a, V3 v) {  // A v
  return v3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
      a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
__device__ __forceinline__ V3 vm(V3 v, const M3& a) {  // row vector v A
  return v3(v.x * a.m[0][0] + v.y * a.m[1][0] + v.z * a.m[2][0], v.x * a.m[0][1] + v.y * a.m[1][1] + v.z * a.m[2][1],
      v.x * a.m[0][2] + v.y * a.m[1][2] + v.z * a.m[2][2]);
}

struct Quat {  // xyzw, unit after load()
  V3 v;
  float w;
};
__device__ __forceinline__ Quat qload(const float* p) {  // the SO3 constructor: normalize(), no epsilon (lie.h:45-47)
  const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
  return Quat{v3(p[0] / n, p[1] / n, p[2] / n), p[3] / n};
}
__device__ __forceinline__ Quat qnorm(Quat q) {
  const float n = sqrtf(dot(q.v, q.v) + q.w * q.w);
  return Quat{v3(q.v.x / n, q.v.y / n, q.v.z / n), q.w / n};
}
__device__ __forceinline__ void qstore(Quat q, float* p) { p[0] = q.v.x, p[1] = q.v.y, p[2] = q.v.z, p[3] = q.w; }
__device__ __forceinline__ Quat qconj(Quat q) { return Quat{-q.v, q.w}; }
__device__ __forceinline__ Quat qmul(Quat a, Quat b) {
  return Quat{v3(a.w * b.v.x + a.v.x * b.w + a.v.y * b.v.z - a.v.z * b.v.y, a.w * b.v.y + a.v.y * b.w + a.v.z * b.v.x - a.v.x * b.v.z,
                  a.w * b.v.z + a.v.z * b.w + a.v.x * b.v.y - a.v.y * b.v.x),
      a.w * b.w - a.v.x * b.v.x - a.v.y * b.v.y - a.v.z * b.v.z};
}
__device__ __forceinline__ V3 qrot(Quat q, V3 p) {
  V3 uv = cross(q.v, p);
  uv = uv + uv;
  return p + q.w * uv + cross(q.v, uv);
}
__device__ __forceinline__ M3 qmat(Quat q) {  // Eigen toRotationMatrix
  const float tx = q.v.x + q.v.x, ty = q.v.y + q.v.y, tz = q.v.z + q.v.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.v.x, txy = ty * q.v.x, txz = tz * q.v.x, tyy = ty * q.v.y, tyz = tz * q.v.y, tzz = tz * q.v.z;
  M3 r = {{{1.f - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1.f - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1.f - (txx + tyy)}}};
  return r;
}
__device__ __forceinline__ Quat so3_exp(V3 phi) {
  const float theta2 = dot(phi, phi), theta = sqrtf(theta2);
  float imag, real;
  if (theta < EPS) {
    const float theta4 = theta2 * theta2;
    imag = 0.5f - (1.0f / 48.0f) * theta2 + (1.0f / 3840.0f) * theta4;
    real = 1.0f - (1.0f / 8.0f) * theta2 + (1.0f / 384.0f) * theta4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  return qnorm(Quat{imag * phi, real});
}
__device__ __forceinline__ V3 so3_log(Quat q) {
  const float n2 = dot(q.v, q.v), w = q.w;
  float f;
  if (n2 < EPS * EPS) {
    f = 2.0f / w - (2.0f / 3.0f) * n2 / (w * w * w);
  } else {
    const float n = sqrtf(n2);
    if (fabsf(w) < EPS)
      f = (w > 0.f ? PI_F : -PI_F) / n;
    else
      f = 2.0f * atanf(n / w) / n;
  }
  return f * q.v;
}
__device__ __forceinline__ M3 so3_left_jacobian(V3 phi) {
  const M3 Phi = hat(phi);
  const float theta2 = dot(phi, phi), theta = sqrtf(theta2);
  const float c1 = theta < EPS ? 0.5f - (1.0f / 24.0f) * theta2 : (1.0f - cosf(theta)) / theta2;
  const float c2 = theta < EPS ? 1.0f / 6.0f - (1.0f / 120.0f) * theta2 : (theta - sinf(theta)) / (theta2 * theta);
  return add(add(eye(), Phi, c1), mm(Phi, Phi), c2);
}
__device__ __forceinline__ M3 so3_left_jacobian_inverse(V3 phi) {
  const M3 Phi = hat(phi);
  const float theta = sqrtf(dot(phi, phi)), half = 0.5f * theta;
  const float c2 = theta < EPS ? 1.0f / 12.0f : (1.0f - theta * cosf(half) / (2.0f * sinf(half))) / (theta * theta);
  return add(add(eye(), Phi, -0.5f), mm(Phi, Phi), c2);
}
__device__ __forceinline__ M3 se3_calcQ(V3 tau, V3 phi) {
  const M3 Tau = hat(tau), Phi = hat(phi);
  const float theta = sqrtf(dot(phi, phi)), t2 = theta * theta, t4 = t2 * t2;
  const float c1 = theta < EPS ? 1.0f / 6.0f - (1.0f / 120.0f) * t2 : (theta - sinf(theta)) / (t2 * theta);
  const float c2 = theta < EPS ? 1.0f / 24.0f - (1.0f / 720.0f) * t2 : (t2 + 2.f * cosf(theta) - 2.f) / (2.f * t4);
  const float c3 = theta < EPS ? 1.0f / 120.0f - (1.0f / 2520.0f) * t2 : (2.f * theta - 3.f * sinf(theta) + theta * cosf(theta)) / (2.f * t4 * theta);
  const M3 PT = mm(Phi, Tau), TP = mm(Tau, Phi), PTP = mm(PT, Phi);
  M3 q = scale(0.5f, Tau);
  q = add(q, add(add(PT, TP), PTP), c1);
  q = add(q, add(add(mm(Phi, PT), mm(TP, Phi)), PTP, -3.f), c2);
  q = add(q, add(mm(PTP, Phi), mm(Phi, PTP)), c3);
  return q;
}
// the non-zero 4 x 3 block J_q of SO3's orthogonal_projector (lie.h:82-90): rows 0..2 = (w I - hat(v)) / 2, row 3 = -v / 2
__device__ __forceinline__ V3 jq_rowvec(Quat q, const float* g4) {  // g4 (1x4) J_q -> 1x3
  const V3 g = v3(g4[0], g4[1], g4[2]);
  // g (w I - hat(v)) / 2 = (w g - g hat(v)) / 2, and g hat(v) = -(hat(v) g)^T = -(v x g) ... row-vector: (g hat(v))_j = sum_i g_i hat(v)_ij = (g x v)_j
  const V3 gh = cross(g, q.v);
  return 0.5f * (q.w * g - gh) + (-0.5f * g4[3]) * q.v;
}
__device__ __forceinline__ void jq_colvec(Quat q, V3 a, float* out4) {  // J_q a (4x1)
  const V3 top = 0.5f * (q.w * a - cross(q.v, a));  // (w I - hat(v)) a / 2
  out4[0] = top.x, out4[1] = top.y, out4[2] = top.z, out4[3] = -0.5f * dot(q.v, a);
}

enum Op { EXP = 0, LOG = 1, INV = 2, MUL = 3, ADJ = 4, ADJT = 5, ACT = 6, ACT4 = 7, TOVEC = 8, FROMVEC = 9, N_OPS = 10 };

// ---- SO3 (group id 1: K = 3, N = 4) -------------------------------------------------------------------------------------------
struct SO3 {
  static constexpr int K = 3, N = 4;
  Quat q;
  __device__ static SO3 load(const float* p) { return SO3{qload(p)}; }
  __device__ void store(float* p) const { qstore(q, p); }
  __device__ static SO3 Exp(const float* a) { return SO3{so3_exp(v3(a[0], a[1], a[2]))}; }
  __device__ void Log(float* a) const {
    const V3 l = so3_log(q);
    a[0] = l.x, a[1] = l.y, a[2] = l.z;
  }
  __device__ SO3 inv() const { return SO3{qnorm(qconj(q))}; }
  __device__ SO3 mul(const SO3& o) const { return SO3{qnorm(qmul(q, o.q))}; }
  __device__ V3 act(V3 p) const { return qrot(q, p); }
  __device__ void act4(const float* p, float* o) const {
    const V3 y = qrot(q, v3(p[0], p[1], p[2]));
    o[0] = y.x, o[1] = y.y, o[2] = y.z, o[3] = p[3];
  }
  // row vector (K) times Adj(X), Adj(X) times column vector
  __device__ void row_Adj(const float* g, float* o) const {
    const V3 r = vm(v3(g[0], g[1], g[2]), qmat(q));
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ void Adj_col(const float* a, float* o) const {
    const V3 r = mv(qmat(q), v3(a[0], a[1], a[2]));
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ void AdjT_col(const float* a, float* o) const { row_Adj(a, o); }  // Adj^T a = (a^T Adj)^T
  __device__ static void row_adj(const float* g, const float* b, float* o) {   // g adj(b), adj(b) = hat(b)
    const V3 r = cross(v3(g[0], g[1], g[2]), v3(b[0], b[1], b[2]));           // (g hat(b))_j = (g x b)_j
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ static void row_left_jacobian(const float* g, const float* a, float* o, bool inverse) {
    const V3 phi = v3(a[0], a[1], a[2]);
    const V3 r = vm(v3(g[0], g[1], g[2]), inverse ? so3_left_jacobian_inverse(phi) : so3_left_jacobian(phi));
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ static void row_act_jacobian(V3 g, V3 y, float* o) {  // g hat(-y) = y x g
    const V3 r = cross(y, g);
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ static void row_act4_jacobian(const float* g4, const float* y4, float* o) { row_act_jacobian(v3(g4[0], g4[1], g4[2]), v3(y4[0], y4[1], y4[2]), o); }
  __device__ V3 row_R(V3 g) const { return vm(g, qmat(q)); }
  __device__ void row_matrix4(const float* g4, float* o) const {
    const V3 r = vm(v3(g4[0], g4[1], g4[2]), qmat(q));
    o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = g4[3];
  }
  __device__ void to_tangent(const float* g, float* o) const {  // g (1x4) J
    const V3 r = jq_rowvec(q, g);
    o[0] = r.x, o[1] = r.y, o[2] = r.z;
  }
  __device__ void from_tangent(const float* g, float* o) const {  // g (1x3) pinv(J) = 4 J_q g^T
    float t[4];
    jq_colvec(q, v3(g[0], g[1], g[2]), t);
    o[0] = 4.f * t[0], o[1] = 4.f * t[1], o[2] = 4.f * t[2], o[3] = 4.f * t[3];
  }
};

// ---- SE3 (group id 3: K = 6 = (tau, phi), N = 7 = (t, q)) ---------------------------------------------------------------------
struct SE3 {
  static constexpr int K = 6, N = 7;
  V3 t;
  Quat q;
  __device__ static SE3 load(const float* p) { return SE3{v3(p[0], p[1], p[2]), qload(p + 3)}; }
  __device__ void store(float* p) const {
    p[0] = t.x, p[1] = t.y, p[2] = t.z;
    qstore(q, p + 3);
  }
  __device__ static SE3 Exp(const float* a) {
    const V3 tau = v3(a[0], a[1], a[2]), phi = v3(a[3], a[4], a[5]);
    return SE3{mv(so3_left_jacobian(phi), tau), so3_exp(phi)};
  }
  __device__ void Log(float* a) const {
    const V3 phi = so3_log(q);
    const V3 tau = mv(so3_left_jacobian_inverse(phi), t);
    a[0] = tau.x, a[1] = tau.y, a[2] = tau.z, a[3] = phi.x, a[4] = phi.y, a[5] = phi.z;
  }
  __device__ SE3 inv() const {
    const Quat qi = qnorm(qconj(q));
    return SE3{-qrot(qi, t), qi};
  }
  __device__ SE3 mul(const SE3& o) const { return SE3{t + qrot(q, o.t), qnorm(qmul(q, o.q))}; }
  __device__ V3 act(V3 p) const { return qrot(q, p) + t; }
  __device__ void act4(const float* p, float* o) const {
    const V3 y = qrot(q, v3(p[0], p[1], p[2])) + p[3] * t;
    o[0] = y.x, o[1] = y.y, o[2] = y.z, o[3] = p[3];
  }
  // Adj = [[R, hat(t) R], [0, R]]
  __device__ void row_Adj(const float* g, float* o) const {  // (a, b) Adj = (a R, a hat(t) R + b R)
    const M3 R = qmat(q);
    const V3 a = v3(g[0], g[1], g[2]), b = v3(g[3], g[4], g[5]);
    const V3 r0 = vm(a, R), r1 = vm(cross(a, t) + b, R);  // a hat(t) = a x t
    o[0] = r0.x, o[1] = r0.y, o[2] = r0.z, o[3] = r1.x, o[4] = r1.y, o[5] = r1.z;
  }
  __device__ void Adj_col(const float* a6, float* o) const {  // Adj (u; w) = (R u + hat(t) R w; R w)
    const M3 R = qmat(q);
    const V3 Ru = mv(R, v3(a6[0], a6[1], a6[2])), Rw = mv(R, v3(a6[3], a6[4], a6[5]));
    const V3 r0 = Ru + cross(t, Rw);
    o[0] = r0.x, o[1] = r0.y, o[2] = r0.z, o[3] = Rw.x, o[4] = Rw.y, o[5] = Rw.z;
  }
  __device__ void AdjT_col(const float* a6, float* o) const { row_Adj(a6, o); }
  __device__ static void row_adj(const float* g, const float* b6, float* o) {  // (a, b) [[Phi, Tau], [0, Phi]] = (a Phi, a Tau + b Phi)
    const V3 a = v3(g[0], g[1], g[2]), b = v3(g[3], g[4], g[5]), tau = v3(b6[0], b6[1], b6[2]), phi = v3(b6[3], b6[4], b6[5]);
    const V3 r0 = cross(a, phi), r1 = cross(a, tau) + cross(b, phi);
    o[0] = r0.x, o[1] = r0.y, o[2] = r0.z, o[3] = r1.x, o[4] = r1.y, o[5] = r1.z;
  }
  __device__ static void row_left_jacobian(const float* g, const float* a6, float* o, bool inverse) {
    const V3 tau = v3(a6[0], a6[1], a6[2]), phi = v3(a6[3], a6[4], a6[5]);
    const V3 ga = v3(g[0], g[1], g[2]), gb = v3(g[3], g[4], g[5]);
    const M3 Q = se3_calcQ(tau, phi);
    V3 r0, r1;
    if (!inverse) {  // [[J, Q], [0, J]]
      const M3 J = so3_left_jacobian(phi);
      r0 = vm(ga, J), r1 = vm(ga, Q) + vm(gb, J);
    } else {  // [[Ji, -Ji Q Ji], [0, Ji]]
      const M3 Ji = so3_left_jacobian_inverse(phi);
      r0 = vm(ga, Ji), r1 = vm(gb, Ji) - vm(vm(vm(ga, Ji), Q), Ji);
    }
    o[0] = r0.x, o[1] = r0.y, o[2] = r0.z, o[3] = r1.x, o[4] = r1.y, o[5] = r1.z;
  }
  __device__ static void row_act_jacobian(V3 g, V3 y, float* o) {  // g [I | hat(-y)]
    const V3 r = cross(y, g);
    o[0] = g.x, o[1] = g.y, o[2] = g.z, o[3] = r.x, o[4] = r.y, o[5] = r.z;
  }
  __device__ static void row_act4_jacobian(const float* g4, const float* y4, float* o) {  // g [[y3 I | hat(-y)], [0]]
    const V3 g = v3(g4[0], g4[1], g4[2]);
    const V3 r = cross(v3(y4[0], y4[1], y4[2]), g);
    o[0] = y4[3] * g.x, o[1] = y4[3] * g.y, o[2] = y4[3] * g.z, o[3] = r.x, o[4] = r.y, o[5] = r.z;
  }
  __device__ V3 row_R(V3 g) const { return vm(g, qmat(q)); }
  __device__ void row_matrix4(const float* g4, float* o) const {  // g [[R, t], [0, 1]]
    const V3 g = v3(g4[0], g4[1], g4[2]);
    const V3 r = vm(g, qmat(q));
    o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = dot(g, t) + g4[3];
  }
  __device__ void to_tangent(const float* g, float* o) const {  // g (1x7) [[I, hat(-t)], [0, J_q]]
    const V3 gt = v3(g[0], g[1], g[2]);
    const V3 phi = cross(t, gt) + jq_rowvec(q, g + 3);  // gt hat(-t) = t x gt
    o[0] = gt.x, o[1] = gt.y, o[2] = gt.z, o[3] = phi.x, o[4] = phi.y, o[5] = phi.z;
  }
  __device__ void from_tangent(const float* g, float* o) const {  // (tau, 4 J_q (phi - tau hat(-t)))
    const V3 tau = v3(g[0], g[1], g[2]), phi = v3(g[3], g[4], g[5]);
    float q4[4];
    jq_colvec(q, phi - cross(t, tau), q4);
    o[0] = tau.x, o[1] = tau.y, o[2] = tau.z, o[3] = 4.f * q4[0], o[4] = 4.f * q4[1], o[5] = 4.f * q4[2], o[6] = 4.f * q4[3];
  }
};

__host__ __device__ inline int x_width(int op, int K, int N) { return op == EXP ? K : N; }
__host__ __device__ inline int y_width(int op, int K, int N) {
  return op == MUL ? N : (op == ADJ || op == ADJT) ? K : op == ACT ? 3 : op == ACT4 ? 4 : 0;
}
__host__ __device__ inline int out_width(int op, int K, int N) {
  return (op == EXP || op == INV || op == MUL || op == TOVEC || op == FROMVEC) ? N : (op == LOG || op == ADJ || op == ADJT) ? K : op == ACT ? 3 : 4;
}

template <class G>
__global__ void __launch_bounds__(256) lie_forward_kernel(int op, long long B, const float* __restrict__ X, const float* __restrict__ Y,
    float* __restrict__ out) {
  const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  constexpr int K = G::K, N = G::N;
  const float* x = X + i * x_width(op, K, N);
  const float* y = Y ? Y + i * y_width(op, K, N) : nullptr;
  float* o = out + i * out_width(op, K, N);
  float xr[7], yr[7];
  for (int c = 0; c < x_width(op, K, N); ++c) xr[c] = x[c];
  for (int c = 0; c < y_width(op, K, N); ++c) yr[c] = y[c];
  switch (op) {
    case EXP: G::Exp(xr).store(o); break;
    case LOG: G::load(xr).Log(o); break;
    case INV: G::load(xr).inv().store(o); break;
    case MUL: G::load(xr).mul(G::load(yr)).store(o); break;
    case ADJ: G::load(xr).Adj_col(yr, o); break;
    case ADJT: G::load(xr).AdjT_col(yr, o); break;
    case ACT: {
      const V3 r = G::load(xr).act(v3(yr[0], yr[1], yr[2]));
      o[0] = r.x, o[1] = r.y, o[2] = r.z;
    } break;
    case ACT4: G::load(xr).act4(yr, o); break;
    default: break;
  }
}

template <class G>
__global__ void __launch_bounds__(256) lie_backward_kernel(int op, long long B, const float* __restrict__ grad, const float* __restrict__ X,
    const float* __restrict__ Y, float* __restrict__ dX, float* __restrict__ dY) {
  const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  constexpr int K = G::K, N = G::N;
  const int xw = x_width(op, K, N), yw = y_width(op, K, N), gw = out_width(op, K, N);
  float xr[7], yr[7], g[7], ox[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, oy[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < xw; ++c) xr[c] = X[i * xw + c];
  for (int c = 0; c < yw; ++c) yr[c] = Y[i * yw + c];
  for (int c = 0; c < gw; ++c) g[c] = grad[i * gw + c];
  switch (op) {
    case EXP: G::row_left_jacobian(g, xr, ox, false); break;  // da = dX[:K] J_l(a)
    case LOG: {                                               // dX = da J_l^-1(log X)
      float a[6];
      G::load(xr).Log(a);
      G::row_left_jacobian(g, a, ox, true);
    } break;
    case INV: {  // dX = -dY Adj(X^-1)
      G::load(xr).inv().row_Adj(g, ox);
      for (int c = 0; c < K; ++c) ox[c] = -ox[c];
    } break;
    case MUL:  // dX = dZ, dY = dZ Adj(X)
      for (int c = 0; c < K; ++c) ox[c] = g[c];
      G::load(xr).row_Adj(g, oy);
      break;
    case ADJ: {  // da = db Adj(X), dX = -db adj(Adj a)
      const G Xg = G::load(xr);
      float b[6];
      Xg.Adj_col(yr, b);
      Xg.row_Adj(g, oy);
      G::row_adj(g, b, ox);
      for (int c = 0; c < K; ++c) ox[c] = -ox[c];
    } break;
    case ADJT: {  // da = Adj(X) db, dX = -a adj(Adj(X) db)
      const G Xg = G::load(xr);
      Xg.Adj_col(g, oy);
      G::row_adj(yr, oy, ox);
      for (int c = 0; c < K; ++c) ox[c] = -ox[c];
    } break;
    case ACT: {  // dp = dq R, dX = dq act_jacobian(X p)
      const G Xg = G::load(xr);
      const V3 gq = v3(g[0], g[1], g[2]);
      G::row_act_jacobian(gq, Xg.act(v3(yr[0], yr[1], yr[2])), ox);
      const V3 dp = Xg.row_R(gq);
      oy[0] = dp.x, oy[1] = dp.y, oy[2] = dp.z;
    } break;
    case ACT4: {
      const G Xg = G::load(xr);
      float y4[4];
      Xg.act4(yr, y4);
      G::row_act4_jacobian(g, y4, ox);
      Xg.row_matrix4(g, oy);
    } break;
    case TOVEC: G::load(xr).to_tangent(g, ox); break;
    case FROMVEC: G::load(xr).from_tangent(g, ox); break;
    default: break;
  }
  if (dX)
    for (int c = 0; c < xw; ++c) dX[i * xw + c] = ox[c];
  if (dY)
    for (int c = 0; c < yw; ++c) dY[i * yw + c] = oy[c];
}

}  // namespace lie
}  // namespace skgs

using namespace skgs;

extern "C" {

int skgs_lie_forward(int32_t group, int32_t op, int64_t B, const float* X, const float* Y, float* out, skgs_stream_t stream) {
  SKGS_REQUIRE(group == 1 || group == 3, "lie: group must be 1 (SO3) or 3 (SE3), lietorch's ids");
  SKGS_REQUIRE(op >= lie::EXP && op <= lie::ACT4, "lie forward: op must be exp .. act4 (vec / InitFromVec are the identity forward)");
  SKGS_REQUIRE(B >= 0, "lie: negative batch");
  if (B == 0) return 0;
  const int K = group == 1 ? 3 : 6, N = group == 1 ? 4 : 7;
  SKGS_REQUIRE(X && out && (lie::y_width(op, K, N) == 0 || Y), "lie forward: NULL argument");
  const dim3 grid((unsigned) ((B + 255) / 256)), block(256);
  if (group == 1)
    hipLaunchKernelGGL(lie::lie_forward_kernel<lie::SO3>, grid, block, 0, (hipStream_t) stream, op, (long long) B, X, Y, out);
  else
    hipLaunchKernelGGL(lie::lie_forward_kernel<lie::SE3>, grid, block, 0, (hipStream_t) stream, op, (long long) B, X, Y, out);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_lie_backward(int32_t group, int32_t op, int64_t B, const float* grad, const float* X, const float* Y, float* dX, float* dY,
    skgs_stream_t stream) {
  SKGS_REQUIRE(group == 1 || group == 3, "lie: group must be 1 (SO3) or 3 (SE3), lietorch's ids");
  SKGS_REQUIRE(op >= lie::EXP && op < lie::N_OPS, "lie backward: unknown op");
  SKGS_REQUIRE(B >= 0, "lie: negative batch");
  if (B == 0) return 0;
  const int K = group == 1 ? 3 : 6, N = group == 1 ? 4 : 7;
  SKGS_REQUIRE(grad && X && (lie::y_width(op, K, N) == 0 || Y) && (dX || dY), "lie backward: NULL argument");
  const dim3 grid((unsigned) ((B + 255) / 256)), block(256);
  if (group == 1)
    hipLaunchKernelGGL(lie::lie_backward_kernel<lie::SO3>, grid, block, 0, (hipStream_t) stream, op, (long long) B, grad, X, Y, dX, dY);
  else
    hipLaunchKernelGGL(lie::lie_backward_kernel<lie::SE3>, grid, block, 0, (hipStream_t) stream, op, (long long) B, grad, X, Y, dX, dY);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
