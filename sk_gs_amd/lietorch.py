"""Stand-in for ``lietorch`` (princeton-vl/lietorch, un-pinned in the reference's requirements_git.txt:2) -- the SO3 / SE3 subset
the reference's deform uses, so that the UNMODIFIED ``networks/sk_gs.py`` runs on a machine without that CUDA-only package::

    import sk_gs_amd
    sk_gs_amd.install_as_lietorch()           # sys.modules['lietorch'] = this module;  `from lietorch import SE3, SO3` (sk_gs.py:12)

What the reference calls (sk_gs.py:193-206, 416-418, 798-828, 875-1027, 1086-1125, 1277-1300, 1371-1381, 1446-1513;
gaussian_splatting.py:160; GS_utils.py:46): ``SE3.InitFromVec``, ``SE3.exp``, ``SO3.InitFromVec``, ``SO3.exp``, ``.vec()``,
``.act(p)``, ``.inv()``, ``*``, ``.log()``, ``.matrix()``, ``[index]``, ``.shape``, broadcasting of same-rank operands.  All of
lietorch's group interface for these two groups is here (``Identity``, ``Random``, ``retr``, ``adj``, ``adjT``, ``Jinv``,
``translation``, ``quaternion``, ``view``, ``detach``, ``to`` ... included).

Semantics follow the vendored copy of lietorch's C++ core in the reference (the authoritative in-repo statement of it):

* forward maps   my_ext/_C/include/lie.h:45-64 (SO3: the constructor NORMALISES the quaternion; act = p + w uv + q x uv),
                 :107-176 (Log / Exp / left Jacobians), :212-252 (SE3 ctor, inv, product, act), :254-263 (Adj), :314-385;
* backward maps  my_ext/_C/src/ops_3d/lie_cpu.cpp:25-236 -- the gradient of a GROUP ELEMENT is the LEFT-TANGENT row vector stored
                 in the first K of its N slots (SO3: K = 3, N = 4; SE3: K = 6 = (tau, phi), N = 7 = (t, q_xyzw)):
                 exp: da = dX J_l(a); log: dX = da J_l^-1(log X); inv: dX = -dY Adj(Y); mul: dX = dZ, dY = dZ Adj(X);
                 act: dp = dq R, dX = dq [I | -hat(X p)];
* embedding <-> tangent at ``InitFromVec`` / ``vec()``: upstream's Python glue (lietorch/group_ops.py ``FromVec`` / ``ToVec``)
  multiplies by pinv(J) / J with J = ``orthogonal_projector`` (lie.h:82-90, 303-311).  That glue is NOT in the reference tree
  (parity unpinned); it is restated here from the published source.  pinv(J) is evaluated in closed form
  (J_q^T J_q = I/4 for the unit quaternion the constructor produces): pinv = [[I, -4 hat(-t) J_q^T], [0, 4 J_q^T]] -- checked
  against ``torch.linalg.pinv`` in tests/test_lietorch_standin.py.

Every op is an ``autograd.Function`` with a pure-torch body (any device and dtype: what runs on CPU) and, for fp32 rows on a HIP
device, ONE launch of libskgs_hip.so per direction (``skgs_lie_forward`` / ``skgs_lie_backward``, csrc/lie_ops.hip -- lietorch's
backend; the torch bodies are ~40-60 small kernels per operator, which made the 20-bone chain of stage `sk` 30 ms of launches per
step).  ``SKGS_LIE_HIP_OPS=0`` keeps the torch bodies.  The one P-sized pattern of the per-frame path,

    (sk_T[indices].act(points[:, None]) * weights[..., None]).sum(dim=1)          sk_gs.py:1147, 814, 1478
    spT[self.p2sp].act(points)                                                     sk_gs.py:816, 1481

is recognised WITHOUT touching the reference: ``group[LongTensor]`` returns a group whose gather is deferred, its ``act`` on a
matching point tensor returns a tensor stand-in (a ``torch.Tensor`` subclass with the right shape / dtype / device) that turns
``* weights[..., None]`` followed by ``.sum(dim=1)`` into ONE launch of ``skgs_se3_blend_forward`` (csrc/lie_blend.hip; backward
``skgs_se3_blend_backward``, tangent-space gradients as above) and materialises the real [P,K,3] tensor through the generic ops
for anything else that is done with it.  On a HIP device the launch is libskgs_hip.so's; there is no silent fallback: a missing
library raises.  ``SKGS_LIE_FUSED=0`` switches the recognition off (generic ops only).
"""
from __future__ import annotations

import ctypes as C
import os
from functools import reduce
from typing import Optional

import torch
from torch import Tensor

__all__ = ['SO3', 'SE3', 'LieGroup', 'LieGroupParameter', 'cat', 'stack']

EPS = 1e-6  # lie.h:23


# ------------------------------------------------------------------------------------------------ small batched helpers
def _hat(v: Tensor) -> Tensor:
    """[..., 3] -> [..., 3, 3]   (lie.h:98-103)"""
    z = torch.zeros_like(v[..., 0])
    return torch.stack([z, -v[..., 2], v[..., 1], v[..., 2], z, -v[..., 0], -v[..., 1], v[..., 0], z], dim=-1).reshape(*v.shape[:-1], 3, 3)


def _cross(a: Tensor, b: Tensor) -> Tensor:
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


def _qnormalize(q: Tensor) -> Tensor:
    """the SO3 constructor (lie.h:45-47): Eigen's normalize(), no epsilon"""
    return q / q.square().sum(dim=-1, keepdim=True).sqrt()


def _qmul(a: Tensor, b: Tensor) -> Tensor:
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qconj(q: Tensor) -> Tensor:
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def _qrot(q: Tensor, p: Tensor) -> Tensor:
    """lie.h:59-64: uv = 2 q.vec x p;  p + w uv + q.vec x uv"""
    uv = _cross(q[..., :3], p)
    uv = uv + uv
    return p + q[..., 3:] * uv + _cross(q[..., :3], uv)


def _qmat(q: Tensor) -> Tensor:
    """Eigen's toRotationMatrix of a unit quaternion, [..., 3, 3]"""
    x, y, z, w = q.unbind(-1)
    tx, ty, tz = x + x, y + y, z + z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return torch.stack([1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx,
                        1 - (txx + tyy)], dim=-1).reshape(*q.shape[:-1], 3, 3)


def _rowvec(g: Tensor, A: Tensor) -> Tensor:
    """row vector times matrix, batched: [..., n] x [..., n, m] -> [..., m]"""
    return torch.matmul(g.unsqueeze(-2), A).squeeze(-2)


def _pad_last(g: Tensor, n: int) -> Tensor:
    """gradient buffers of group elements are N wide with the tangent in the first K slots (lie_cpu.cpp:371-489: zeros(X.sizes()))"""
    return torch.cat([g, g.new_zeros(*g.shape[:-1], n - g.shape[-1])], dim=-1) if g.shape[-1] < n else g


# ------------------------------------------------------------------------------------------------ SO3 maths (lie.h:27-210)
def _so3_exp(phi: Tensor) -> Tensor:
    theta2 = phi.square().sum(dim=-1, keepdim=True)
    theta = theta2.sqrt()
    small = theta < EPS
    theta4 = theta2 * theta2
    safe = torch.where(small, torch.ones_like(theta), theta)
    imag = torch.where(small, 0.5 - (1.0 / 48.0) * theta2 + (1.0 / 3840.0) * theta4, torch.sin(0.5 * safe) / safe)
    real = torch.where(small, 1.0 - (1.0 / 8.0) * theta2 + (1.0 / 384.0) * theta4, torch.cos(0.5 * safe))
    return _qnormalize(torch.cat([imag * phi, real], dim=-1))


def _so3_log(q: Tensor) -> Tensor:
    v, w = q[..., :3], q[..., 3:]
    n2 = v.square().sum(dim=-1, keepdim=True)
    n = n2.sqrt()
    tiny = n2 < EPS * EPS
    w_small = w.abs() < EPS
    safe_n = torch.where(tiny, torch.ones_like(n), n)
    safe_w = torch.where(w_small & ~tiny, torch.ones_like(w), w)
    series = 2.0 / safe_w - (2.0 / 3.0) * n2 / (safe_w * safe_w * safe_w)
    at_pi = torch.where(w > 0, torch.pi / safe_n, -torch.pi / safe_n)
    general = 2.0 * torch.atan(safe_n / safe_w) / safe_n
    return torch.where(tiny, series, torch.where(w_small, at_pi, general)) * v


def _so3_left_jacobian(phi: Tensor) -> Tensor:
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    Phi = _hat(phi)
    theta2 = phi.square().sum(dim=-1)[..., None, None]
    theta = theta2.sqrt()
    small = theta < EPS
    safe2 = torch.where(small, torch.ones_like(theta2), theta2)
    safe = torch.where(small, torch.ones_like(theta), theta)
    c1 = torch.where(small, 0.5 - (1.0 / 24.0) * theta2, (1.0 - torch.cos(safe)) / safe2)
    c2 = torch.where(small, 1.0 / 6.0 - (1.0 / 120.0) * theta2, (safe - torch.sin(safe)) / (safe2 * safe))
    return I + c1 * Phi + c2 * (Phi @ Phi)


def _so3_left_jacobian_inverse(phi: Tensor) -> Tensor:
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    Phi = _hat(phi)
    theta = phi.square().sum(dim=-1).sqrt()[..., None, None]
    small = theta < EPS
    safe = torch.where(small, torch.ones_like(theta), theta)
    half = 0.5 * safe
    c2 = torch.where(small, torch.full_like(theta, 1.0 / 12.0), (1.0 - safe * torch.cos(half) / (2.0 * torch.sin(half))) / (safe * safe))
    return I - 0.5 * Phi + c2 * (Phi @ Phi)


def _so3_projector(q: Tensor) -> Tensor:
    """the non-zero 4 x 3 block of orthogonal_projector (lie.h:82-90): rows 0..2 = (w I + hat(-v)) / 2, row 3 = -v / 2"""
    v, w = q[..., :3], q[..., 3]
    I = torch.eye(3, dtype=q.dtype, device=q.device)
    top = 0.5 * (w[..., None, None] * I + _hat(-v))
    return torch.cat([top, (-0.5 * v).unsqueeze(-2)], dim=-2)


# ------------------------------------------------------------------------------------------------ SE3 maths (lie.h:212-393)
def _se3_split(X: Tensor):
    return X[..., :3], _qnormalize(X[..., 3:7])


def _se3_adj(t: Tensor, q: Tensor) -> Tensor:
    """[[R, hat(t) R], [0, R]]   (lie.h:254-263)"""
    R = _qmat(q)
    top = torch.cat([R, _hat(t) @ R], dim=-1)
    bot = torch.cat([torch.zeros_like(R), R], dim=-1)
    return torch.cat([top, bot], dim=-2)


def _se3_calcQ(tau: Tensor, phi: Tensor) -> Tensor:
    Tau, Phi = _hat(tau), _hat(phi)
    theta = phi.square().sum(dim=-1).sqrt()[..., None, None]
    t2 = theta * theta
    t4 = t2 * t2
    small = theta < EPS
    s = torch.where(small, torch.ones_like(theta), theta)
    s2, s4 = s * s, s * s * s * s
    c1 = torch.where(small, 1.0 / 6.0 - (1.0 / 120.0) * t2, (s - torch.sin(s)) / (s2 * s))
    c2 = torch.where(small, 1.0 / 24.0 - (1.0 / 720.0) * t2, (s2 + 2 * torch.cos(s) - 2) / (2 * s4))
    c3 = torch.where(small, 1.0 / 120.0 - (1.0 / 2520.0) * t2, (2 * s - 3 * torch.sin(s) + s * torch.cos(s)) / (2 * s4 * s))
    PT, TP = Phi @ Tau, Tau @ Phi
    PTP = PT @ Phi
    return (0.5 * Tau + c1 * (PT + TP + PTP) + c2 * (Phi @ PT + TP @ Phi - 3 * PTP) + c3 * (PTP @ Phi + Phi @ PTP))


def _se3_left_jacobian(a: Tensor) -> Tensor:
    tau, phi = a[..., :3], a[..., 3:6]
    J, Q = _so3_left_jacobian(phi), _se3_calcQ(tau, phi)
    return torch.cat([torch.cat([J, Q], dim=-1), torch.cat([torch.zeros_like(J), J], dim=-1)], dim=-2)


def _se3_left_jacobian_inverse(a: Tensor) -> Tensor:
    tau, phi = a[..., :3], a[..., 3:6]
    Ji, Q = _so3_left_jacobian_inverse(phi), _se3_calcQ(tau, phi)
    return torch.cat([torch.cat([Ji, -Ji @ Q @ Ji], dim=-1), torch.cat([torch.zeros_like(Ji), Ji], dim=-1)], dim=-2)


def _se3_log(X: Tensor) -> Tensor:
    t, q = _se3_split(X)
    phi = _so3_log(q)
    return torch.cat([(_so3_left_jacobian_inverse(phi) @ t.unsqueeze(-1)).squeeze(-1), phi], dim=-1)


def _se3_exp(a: Tensor) -> Tensor:
    tau, phi = a[..., :3], a[..., 3:6]
    return torch.cat([(_so3_left_jacobian(phi) @ tau.unsqueeze(-1)).squeeze(-1), _so3_exp(phi)], dim=-1)


class _SO3Math:
    group_id, K, N = 1, 3, 4

    @staticmethod
    def canon(X):
        return _qnormalize(X)

    exp, log = staticmethod(_so3_exp), staticmethod(lambda X: _so3_log(_qnormalize(X)))
    left_jacobian, left_jacobian_inverse = staticmethod(_so3_left_jacobian), staticmethod(_so3_left_jacobian_inverse)

    @staticmethod
    def inv(X):
        return _qnormalize(_qconj(_qnormalize(X)))

    @staticmethod
    def mul(X, Y):
        return _qnormalize(_qmul(_qnormalize(X), _qnormalize(Y)))

    @staticmethod
    def act(X, p):
        return _qrot(_qnormalize(X), p)

    @staticmethod
    def act4(X, p):
        return torch.cat([_qrot(_qnormalize(X), p[..., :3]), p[..., 3:]], dim=-1)

    @staticmethod
    def Adj(X):
        return _qmat(_qnormalize(X))

    @staticmethod
    def adj(a):
        return _hat(a)

    @staticmethod
    def rotation(X):
        return _qmat(_qnormalize(X))

    @staticmethod
    def matrix4(X):
        R = _qmat(_qnormalize(X))
        T = torch.zeros(*R.shape[:-2], 4, 4, dtype=R.dtype, device=R.device)
        T[..., :3, :3] = R
        T[..., 3, 3] = 1
        return T

    @staticmethod
    def act_jacobian(y):
        return _hat(-y)

    @staticmethod
    def act4_jacobian(y):
        J = _hat(-y[..., :3])
        return torch.cat([J, torch.zeros_like(J[..., :1, :])], dim=-2)

    @staticmethod
    def to_tangent(X, g):
        """ToVec backward: g J"""
        return _pad_last(_rowvec(g, _so3_projector(_qnormalize(X))), 4)

    @staticmethod
    def from_tangent(X, g):
        """FromVec backward: g pinv(J), pinv(J_q) = 4 J_q^T"""
        return 4.0 * _rowvec(g[..., :3], _so3_projector(_qnormalize(X)).transpose(-1, -2))

    @staticmethod
    def projector(X):
        J = _so3_projector(_qnormalize(X))
        return torch.cat([J, torch.zeros_like(J[..., :1])], dim=-1)


class _SE3Math:
    group_id, K, N = 3, 6, 7

    @staticmethod
    def canon(X):
        return torch.cat([X[..., :3], _qnormalize(X[..., 3:7])], dim=-1)

    exp, log = staticmethod(_se3_exp), staticmethod(_se3_log)
    left_jacobian, left_jacobian_inverse = staticmethod(_se3_left_jacobian), staticmethod(_se3_left_jacobian_inverse)

    @staticmethod
    def inv(X):
        t, q = _se3_split(X)
        qi = _qnormalize(_qconj(q))
        return torch.cat([-_qrot(qi, t), qi], dim=-1)

    @staticmethod
    def mul(X, Y):
        tx, qx = _se3_split(X)
        ty, qy = _se3_split(Y)
        return torch.cat([tx + _qrot(qx, ty), _qnormalize(_qmul(qx, qy))], dim=-1)

    @staticmethod
    def act(X, p):
        t, q = _se3_split(X)
        return _qrot(q, p) + t

    @staticmethod
    def act4(X, p):
        t, q = _se3_split(X)
        return torch.cat([_qrot(q, p[..., :3]) + t * p[..., 3:], p[..., 3:]], dim=-1)

    @staticmethod
    def Adj(X):
        return _se3_adj(*_se3_split(X))

    @staticmethod
    def adj(a):
        Tau, Phi = _hat(a[..., :3]), _hat(a[..., 3:6])
        return torch.cat([torch.cat([Phi, Tau], dim=-1), torch.cat([torch.zeros_like(Phi), Phi], dim=-1)], dim=-2)

    @staticmethod
    def rotation(X):
        return _qmat(_qnormalize(X[..., 3:7]))

    @staticmethod
    def matrix4(X):
        t, q = _se3_split(X)
        R = _qmat(q)
        T = torch.zeros(*R.shape[:-2], 4, 4, dtype=R.dtype, device=R.device)
        T[..., :3, :3] = R
        T[..., :3, 3] = t
        T[..., 3, 3] = 1
        return T

    @staticmethod
    def act_jacobian(y):
        I = torch.eye(3, dtype=y.dtype, device=y.device).expand(*y.shape[:-1], 3, 3)
        return torch.cat([I, _hat(-y)], dim=-1)

    @staticmethod
    def act4_jacobian(y):
        I = torch.eye(3, dtype=y.dtype, device=y.device) * y[..., 3, None, None]
        J = torch.cat([I, _hat(-y[..., :3])], dim=-1)
        return torch.cat([J, torch.zeros_like(J[..., :1, :])], dim=-2)

    @staticmethod
    def to_tangent(X, g):
        """ToVec backward: g J, J = [[I, hat(-t)], [0, J_q]]  (lie.h:303-311)"""
        t, q = _se3_split(X)
        phi = _rowvec(g[..., :3], _hat(-t)) + _rowvec(g[..., 3:7], _so3_projector(q))
        return _pad_last(torch.cat([g[..., :3], phi], dim=-1), 7)

    @staticmethod
    def from_tangent(X, g):
        """FromVec backward: g pinv(J) = (tau, 4 J_q (phi - tau hat(-t)))"""
        t, q = _se3_split(X)
        tau, phi = g[..., :3], g[..., 3:6]
        gq = 4.0 * _rowvec(phi - _rowvec(tau, _hat(-t)), _so3_projector(q).transpose(-1, -2))
        return torch.cat([tau, gq], dim=-1)

    @staticmethod
    def projector(X):
        t, q = _se3_split(X)
        J = torch.zeros(*X.shape[:-1], 7, 7, dtype=X.dtype, device=X.device)
        J[..., :3, :3] = torch.eye(3, dtype=X.dtype, device=X.device)
        J[..., :3, 3:6] = _hat(-t)
        J[..., 3:7, 3:6] = _so3_projector(q)
        return J


_MATH = {_SO3Math.group_id: _SO3Math, _SE3Math.group_id: _SE3Math}


# ------------------------------------------------------------------------------------------------ group ops (autograd)
def _on_hip(*ts) -> bool:
    """the operator's rows live on a HIP device in fp32: libskgs_hip.so's launch (skgs_lie_forward / _backward, csrc/lie_ops.hip)
    serves it; anything else (CPU tensors, fp64 -- the tests' finite differences) takes the pure-torch body below"""
    return _HIP_OPS and all(t.is_cuda and t.dtype == torch.float32 for t in ts)


_HIP_OPS = os.environ.get('SKGS_LIE_HIP_OPS', '1') != '0'
hip_op_calls = {'forward': 0, 'backward': 0}  # counters (tests)


def _hip_forward(G, op: int, out_width: int, X: Tensor, Y: Optional[Tensor] = None) -> Tensor:
    from sk_gs_amd import _C
    lib = _C.load_library()
    with _C._on_device(X.device):
        X = X.contiguous()
        Y = None if Y is None else Y.contiguous()
        out = torch.empty((X.shape[0], out_width), dtype=torch.float32, device=X.device)
        _C._check(lib.skgs_lie_forward(C.c_int32(G.group_id), C.c_int32(op), C.c_int64(X.shape[0]), C.c_void_p(_C._ptr(X)), C.c_void_p(_C._ptr(Y)),
                                       C.c_void_p(_C._ptr(out)), _C._stream()))
    hip_op_calls['forward'] += 1
    return out


def _hip_backward(G, op: int, grad: Tensor, X: Tensor, Y: Optional[Tensor] = None, need=(True, True)):
    from sk_gs_amd import _C
    lib = _C.load_library()
    with _C._on_device(X.device):
        grad, X = grad.contiguous(), X.contiguous()
        Y = None if Y is None else Y.contiguous()
        dX = torch.empty_like(X) if need[0] else None
        dY = torch.empty_like(Y) if (Y is not None and need[1]) else None
        if dX is not None or dY is not None:
            _C._check(lib.skgs_lie_backward(C.c_int32(G.group_id), C.c_int32(op), C.c_int64(X.shape[0]), C.c_void_p(_C._ptr(grad)),
                                            C.c_void_p(_C._ptr(X)), C.c_void_p(_C._ptr(Y)), C.c_void_p(_C._ptr(dX)), C.c_void_p(_C._ptr(dY)),
                                            _C._stream()))
    hip_op_calls['backward'] += 1
    return (dX,) if Y is None else (dX, dY)


class _GroupOp(torch.autograd.Function):
    """lietorch/group_ops.py GroupOp: forward(group_id, *inputs) on [B, dim] rows; backward hands the row-vector gradients of
    lie_cpu.cpp back (N wide for group elements).  Rows on a HIP device: one launch of csrc/lie_ops.hip per direction (`hip_op`:
    the operator's number there); otherwise the pure-torch ``forward_op`` / ``backward_op``."""
    hip_op = None

    @classmethod
    def out_width(cls, G):
        return G.N

    @classmethod
    def forward(cls, ctx, group_id, *inputs):
        ctx.group_id = group_id
        ctx.save_for_backward(*inputs)
        G = _MATH[group_id]
        if cls.hip_op is not None and _on_hip(*inputs) and inputs[0].shape[0] > 0:
            return _hip_forward(G, cls.hip_op, cls.out_width(G), *inputs)
        return cls.forward_op(G, *inputs)

    @classmethod
    @torch.autograd.function.once_differentiable
    def backward(cls, ctx, grad):
        G = _MATH[ctx.group_id]
        inputs = ctx.saved_tensors
        if cls.hip_op is not None and _on_hip(grad, *inputs) and inputs[0].shape[0] > 0:
            return (None,) + tuple(_hip_backward(G, cls.hip_op, grad, *inputs, need=ctx.needs_input_grad[1:] + (True,)))
        return (None,) + tuple(cls.backward_op(G, grad.contiguous(), *inputs))


class Exp(_GroupOp):
    hip_op = 0

    @staticmethod
    def forward_op(G, a):
        return G.exp(a)

    @staticmethod
    def backward_op(G, grad, a):  # lie_cpu.cpp:25-38
        return (_rowvec(grad[..., :G.K], G.left_jacobian(a)),)


class Log(_GroupOp):
    hip_op = 1

    @classmethod
    def out_width(cls, G):
        return G.K

    @staticmethod
    def forward_op(G, X):
        return G.log(X)

    @staticmethod
    def backward_op(G, grad, X):  # lie_cpu.cpp:54-67
        return (_pad_last(_rowvec(grad, G.left_jacobian_inverse(G.log(X))), G.N),)


class Inv(_GroupOp):
    hip_op = 2

    @staticmethod
    def forward_op(G, X):
        return G.inv(X)

    @staticmethod
    def backward_op(G, grad, X):  # lie_cpu.cpp:84-97
        return (_pad_last(-_rowvec(grad[..., :G.K], G.Adj(G.inv(X))), G.N),)


class Mul(_GroupOp):
    hip_op = 3

    @staticmethod
    def forward_op(G, X, Y):
        return G.mul(X, Y)

    @staticmethod
    def backward_op(G, grad, X, Y):  # lie_cpu.cpp:111-126
        dZ = grad[..., :G.K]
        return _pad_last(dZ, G.N), _pad_last(_rowvec(dZ, G.Adj(X)), G.N)


class Adj(_GroupOp):
    hip_op = 4

    @classmethod
    def out_width(cls, G):
        return G.K

    @staticmethod
    def forward_op(G, X, a):
        return (G.Adj(X) @ a.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def backward_op(G, grad, X, a):  # lie_cpu.cpp:143-162
        A = G.Adj(X)
        b = (A @ a.unsqueeze(-1)).squeeze(-1)
        return _pad_last(-_rowvec(grad, G.adj(b)), G.N), _rowvec(grad, A)


class AdjT(_GroupOp):
    hip_op = 5

    @classmethod
    def out_width(cls, G):
        return G.K

    @staticmethod
    def forward_op(G, X, a):
        return (G.Adj(X).transpose(-1, -2) @ a.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def backward_op(G, grad, X, a):  # lie_cpu.cpp:179-198
        Adb = (G.Adj(X) @ grad.unsqueeze(-1)).squeeze(-1)
        return _pad_last(-_rowvec(a, G.adj(Adb)), G.N), Adb


class Act3(_GroupOp):
    hip_op = 6

    @classmethod
    def out_width(cls, G):
        return 3

    @staticmethod
    def forward_op(G, X, p):
        return G.act(X, p)

    @staticmethod
    def backward_op(G, grad, X, p):  # lie_cpu.cpp:217-236
        y = G.act(X, p)
        return _pad_last(_rowvec(grad, G.act_jacobian(y)), G.N), _rowvec(grad, G.rotation(X))


class Act4(_GroupOp):
    hip_op = 7

    @classmethod
    def out_width(cls, G):
        return 4

    @staticmethod
    def forward_op(G, X, p):
        return G.act4(X, p)

    @staticmethod
    def backward_op(G, grad, X, p):  # lie_cpu.cpp:288-309
        y = G.act4(X, p)
        return _pad_last(_rowvec(grad, G.act4_jacobian(y)), G.N), _rowvec(grad, G.matrix4(X))


class Jinv(_GroupOp):
    """left-Jacobian-inverse action (lie_cpu.cpp:336-350); upstream defines no backward for it"""

    @staticmethod
    def forward_op(G, X, a):
        return (G.left_jacobian_inverse(G.log(X)) @ a.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def backward_op(G, grad, X, a):
        raise AssertionError('Backward operation not implemented for Jinv')


class ToMatrix(_GroupOp):
    """4x4 matrices, forward only (lie_cpu.cpp:312-325); ``LieGroup.matrix()`` is differentiable through Act4 as upstream"""

    @staticmethod
    def forward_op(G, X):
        return G.matrix4(X)

    @staticmethod
    def backward_op(G, grad, X):
        raise AssertionError('Backward operation not implemented for ToMatrix')


class FromVec(torch.autograd.Function):
    """vector -> group element: identity forward; backward = tangent gradient times pinv(orthogonal_projector)"""

    @staticmethod
    def forward(ctx, group_id, a):
        ctx.group_id = group_id
        ctx.save_for_backward(a)
        return a.view_as(a)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        (a,) = ctx.saved_tensors
        if _on_hip(grad, a) and a.shape[0] > 0:
            return None, _hip_backward(_MATH[ctx.group_id], 9, grad, a)[0]
        return None, _MATH[ctx.group_id].from_tangent(a, grad)


class ToVec(torch.autograd.Function):
    """group element -> vector: identity forward; backward = embedding gradient times orthogonal_projector"""

    @staticmethod
    def forward(ctx, group_id, X):
        ctx.group_id = group_id
        ctx.save_for_backward(X)
        return X.view_as(X)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        (X,) = ctx.saved_tensors
        if _on_hip(grad, X) and X.shape[0] > 0:
            return None, _hip_backward(_MATH[ctx.group_id], 8, grad, X)[0]
        return None, _MATH[ctx.group_id].to_tangent(X, grad)


def projector(group_id: int, X: Tensor) -> Tensor:
    """[..., N, N] orthogonal projector (``lie_projector``, lie_torch.cpp:316-324)"""
    return _MATH[group_id].projector(X)


# ------------------------------------------------------------------------------------------------ broadcasting (lietorch/broadcasting.py)
def _check_broadcastable(x: Tensor, y: Tensor):
    assert len(x.shape) == len(y.shape), f'lietorch broadcasting needs operands of the same rank, got {tuple(x.shape)} and {tuple(y.shape)}'
    for n, m in zip(x.shape[:-1], y.shape[:-1]):
        assert n == m or n == 1 or m == 1, f'shapes {tuple(x.shape)} and {tuple(y.shape)} do not broadcast'


def _broadcast_inputs(x: Tensor, y: Optional[Tensor]):
    if y is None:
        return (x.reshape(-1, x.shape[-1]).contiguous(),), tuple(x.shape[:-1])
    _check_broadcastable(x, y)
    xs, xd, ys, yd = x.shape[:-1], x.shape[-1], y.shape[:-1], y.shape[-1]
    out_shape = tuple(max(n, m) for n, m in zip(xs, ys))
    if xs == ys:
        return (x.reshape(-1, xd).contiguous(), y.reshape(-1, yd).contiguous()), out_shape
    x1 = x.expand(*out_shape, xd).reshape(-1, xd).contiguous()
    y1 = y.expand(*out_shape, yd).reshape(-1, yd).contiguous()
    return (x1, y1), out_shape


# ------------------------------------------------------------------------------------------------ the fused skinning expression
_FUSED = os.environ.get('SKGS_LIE_FUSED', '1') != '0'
#: CPU runs of the recognition (tests): the deferred expression is then evaluated by the generic ops in one place
_FUSED_ON_CPU = os.environ.get('SKGS_LIE_FUSED_CPU', '0') == '1'
fused_calls = {'forward': 0, 'backward': 0, 'materialised': 0}  # counters (tests; `sk_gs_amd.lietorch.fused_calls`)


def _blend_reference(T: Tensor, idx: Tensor, points: Tensor, weights: Optional[Tensor]) -> Tensor:
    """the expression through the generic ops, exactly as upstream lietorch would evaluate it: gather, broadcast, act, mul, sum"""
    X = T[idx]  # [P,K,7]
    (x1, p1), out_shape = _broadcast_inputs(X, points[:, None, :])
    y = Act3.apply(_SE3Math.group_id, x1, p1).view(out_shape + (3,))
    if weights is not None:
        y = y * weights[..., None]
    return y.sum(dim=1)


class _SE3Blend(torch.autograd.Function):
    """sum_k w[n,k] * SE3(T[idx[n,k]]).act(p[n]) on a HIP device: skgs_se3_blend_forward / _backward (csrc/lie_blend.hip).
    The gradient handed to ``T`` is lietorch's: tangent rows (tau, phi) in slots 0..5 of [M,7] (what Act3 + the gather's
    index_add would produce)."""

    @staticmethod
    def forward(ctx, T, idx, points, weights):
        from sk_gs_amd import _C
        lib = _C.load_library()
        dev = T.device
        with _C._on_device(dev):
            Tc, pc = _C._f32c(T, dev), _C._f32c(points, dev)
            wc = None if weights is None else _C._f32c(weights, dev)
            ic = idx.contiguous()
            P, K = ic.shape
            M = Tc.shape[0]
            out = torch.empty((P, 3), dtype=torch.float32, device=dev)
            _C._check(lib.skgs_se3_blend_forward(C.c_int32(P), C.c_int32(K), C.c_int32(M), C.c_void_p(_C._ptr(Tc)), C.c_void_p(_C._ptr(ic)),
                                                 C.c_void_p(_C._ptr(pc)), C.c_void_p(_C._ptr(wc)), C.c_void_p(_C._ptr(out)), _C._stream()))
        ctx.save_for_backward(Tc, ic, pc, wc)
        ctx.mark_non_differentiable(idx)
        fused_calls['forward'] += 1
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        from sk_gs_amd import _C
        lib = _C.load_library()
        Tc, ic, pc, wc = ctx.saved_tensors
        dev = Tc.device
        (P, K), M = ic.shape, Tc.shape[0]
        need_T, _, need_p, need_w = ctx.needs_input_grad
        with _C._on_device(dev):
            g = _C._f32c(g_out, dev)
            g_T = torch.empty((M, 7), dtype=torch.float32, device=dev)
            g_w = torch.empty((P, K), dtype=torch.float32, device=dev) if (need_w and wc is not None) else None
            g_p = torch.empty((P, 3), dtype=torch.float32, device=dev) if need_p else None
            lib.skgs_se3_blend_backward_workspace_bytes.restype = C.c_size_t
            nbytes = int(lib.skgs_se3_blend_backward_workspace_bytes(C.c_int32(P), C.c_int32(M)))
            ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            _C._check(lib.skgs_se3_blend_backward(
                C.c_int32(P), C.c_int32(K), C.c_int32(M), C.c_void_p(_C._ptr(Tc)), C.c_void_p(_C._ptr(ic)), C.c_void_p(_C._ptr(pc)),
                C.c_void_p(_C._ptr(wc)), C.c_void_p(_C._ptr(g)), C.c_void_p(_C._ptr(g_T)), C.c_void_p(_C._ptr(g_w)),
                C.c_void_p(_C._ptr(g_p)), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel() * 4), _C._stream()))
        fused_calls['backward'] += 1
        return (g_T if need_T else None), None, g_p, g_w


def _se3_blend(T: Tensor, idx: Tensor, points: Tensor, weights: Optional[Tensor]) -> Tensor:
    if T.is_cuda:
        return _SE3Blend.apply(T, idx, points, weights)
    fused_calls['forward'] += 1
    return _blend_reference(T, idx, points, weights)


class _DeferredAct(torch.Tensor):
    """What ``SE3[LongTensor [P,K]].act(points [P,1,3])`` returns when the fused path applies: a tensor of shape [P,K,3] whose
    values are not computed yet.  ``* w [P,K,1]`` and ``.sum(dim=1)`` are absorbed; any other use computes the values through the
    generic ops first (``fused_calls['materialised']`` counts those) and proceeds with the plain tensor."""

    @staticmethod
    def __new__(cls, T, idx, points, weights=None):
        P, K = idx.shape
        r = torch.Tensor._make_wrapper_subclass(cls, (P, K, 3), dtype=T.dtype, device=T.device, requires_grad=False)
        r._skgs = (T, idx, points, weights)
        return r

    def _materialise(self) -> Tensor:
        T, idx, points, weights = self._skgs
        fused_calls['materialised'] += 1
        X = T[idx]
        (x1, p1), out_shape = _broadcast_inputs(X, points[:, None, :])
        y = Act3.apply(_SE3Math.group_id, x1, p1).view(out_shape + (3,))
        return y if weights is None else y * weights[..., None]

    _MUL = None
    _SUM = None
    _META = {'size', 'dim', 'ndimension', 'numel', 'nelement', 'is_floating_point', 'is_complex', 'get_device', 'element_size'}

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            if cls._MUL is None:
                cls._MUL = {torch.mul, torch.Tensor.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.multiply, torch.Tensor.multiply}
                cls._SUM = {torch.sum, torch.Tensor.sum}
            name = getattr(func, '__name__', '')
            if name == '__get__' or name in cls._META:  # shape / dtype / device / dim() ...: the stand-in carries them itself
                return func(*args, **kwargs)
            if func in cls._MUL and len(args) == 2 and not kwargs:
                a, b = args
                me, other = (a, b) if isinstance(a, cls) else (b, a)
                if (isinstance(me, cls) and isinstance(other, Tensor) and not isinstance(other, cls) and me._skgs[3] is None
                        and tuple(other.shape) == (me.shape[0], me.shape[1], 1) and other.dtype == me.dtype and other.device == me.device):
                    T, idx, points, _ = me._skgs
                    return cls(T, idx, points, other[..., 0])
            if func in cls._SUM and isinstance(args[0], cls):
                dim = kwargs.get('dim', args[1] if len(args) > 1 else None)
                keepdim = kwargs.get('keepdim', args[2] if len(args) > 2 else False)
                extra = set(kwargs) - {'dim', 'keepdim'}
                if dim in (1, -2, (1,), [1]) and not keepdim and not extra:
                    return _se3_blend(*args[0]._skgs)

            def real(x):
                if isinstance(x, cls):
                    return x._materialise()
                if isinstance(x, (list, tuple)):
                    return type(x)(real(e) for e in x)
                return x
            return func(*real(args), **{k: real(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # every Python-level use goes through __torch_function__ above; an op that reaches the dispatcher with the stand-in came
        # from C++ and would compute below autograd: refuse rather than drop a gradient
        raise RuntimeError(f'sk_gs_amd.lietorch: the deferred SE3[indices].act(points) reached {func} without being evaluated; '
                           'set SKGS_LIE_FUSED=0 to use the generic ops and please report the call site')

    def __repr__(self):
        return f'_DeferredAct(shape={tuple(self.shape)}, weighted={self._skgs[3] is not None})'


# ------------------------------------------------------------------------------------------------ the group classes (lietorch/groups.py)
class LieGroup:
    """Base class of SO3 / SE3: ``data`` [..., N] plus lietorch's interface."""
    _math = None
    group_name = None

    def __init__(self, data: Tensor):
        self.data = data

    def _new(self, data: Tensor):
        """a group of this element's kind around `data` (upstream: ``self.__class__(data)``)"""
        return type(self)(data)

    def __repr__(self):
        return '{}: size={}, device={}, dtype={}'.format(self.group_name, self.shape, self.device, self.dtype)

    # -- introspection
    @property
    def group_id(self):
        return self._math.group_id

    @property
    def manifold_dim(self):
        return self._math.K

    @property
    def embedded_dim(self):
        return self._math.N

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (self.manifold_dim,)

    # -- construction
    @classmethod
    def apply_op(cls, op, x, y=None):
        inputs, out_shape = _broadcast_inputs(x, y)
        data = op.apply(cls._math.group_id, *inputs)
        return data.view(out_shape + data.shape[1:])

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if len(batch_shape) and isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        numel = reduce(lambda x, y: x * y, batch_shape, 1)
        data = cls.id_elem.reshape(1, -1)
        if 'device' in kwargs:
            data = data.to(kwargs['device'])
        if 'dtype' in kwargs:
            data = data.type(kwargs['dtype'])
        return cls(data.repeat(numel, 1)).view(tuple(batch_shape))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def InitFromVec(cls, data: Tensor):
        return cls(cls.apply_op(FromVec, data))

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        if len(batch_shape) and isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        return cls.exp(sigma * torch.randn(*(tuple(batch_shape) + (cls._math.K,)), **kwargs))

    @classmethod
    def exp(cls, x: Tensor):
        """exponential map: tangent [..., K] -> group"""
        return cls(cls.apply_op(Exp, x))

    # -- maps
    def vec(self) -> Tensor:
        return self.apply_op(ToVec, self.data)

    def quaternion(self) -> Tensor:
        """the (normalised) rotation quaternion, xyzw"""
        return _qnormalize(self.data[..., -4:])

    def log(self) -> Tensor:
        return self.apply_op(Log, self.data)

    def inv(self):
        return self._new(self.apply_op(Inv, self.data))

    def mul(self, other):
        return self._new(self.apply_op(Mul, self.data, other.data))

    def retr(self, a: Tensor):
        """retraction: Exp(a) * X"""
        dX = self.__class__.apply_op(Exp, a)
        return self._new(self.apply_op(Mul, dX, self.data))

    def adj(self, a: Tensor) -> Tensor:
        return self.apply_op(Adj, self.data, a)

    def adjT(self, a: Tensor) -> Tensor:
        return self.apply_op(AdjT, self.data, a)

    def Jinv(self, a: Tensor) -> Tensor:
        return self.apply_op(Jinv, self.data, a)

    def act(self, p: Tensor) -> Tensor:
        if p.shape[-1] == 3:
            return self.apply_op(Act3, self.data, p)
        if p.shape[-1] == 4:
            return self.apply_op(Act4, self.data, p)
        raise ValueError(f'act: points must be [..., 3] or [..., 4], got {tuple(p.shape)}')

    def matrix(self) -> Tensor:
        """[..., 4, 4]; differentiable (the action on the identity's columns, as upstream)"""
        I = torch.eye(4, dtype=self.dtype, device=self.device)
        I = I.view([1] * (len(self.data.shape) - 1) + [4, 4])
        return self._new(self.data[..., None, :]).act(I).transpose(-1, -2)

    def translation(self) -> Tensor:
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        p = p.view([1] * (len(self.data.shape) - 1) + [4])
        return self.apply_op(Act4, self.data, p)

    # -- tensor-like plumbing
    def detach(self):
        return self._new(self.data.detach())

    def view(self, dims):
        return self._new(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    def __getitem__(self, index):
        return self._new(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def __len__(self):
        return self.data.shape[0]

    def to(self, *args, **kwargs):
        return self._new(self.data.to(*args, **kwargs))

    def cpu(self):
        return self._new(self.data.cpu())

    def cuda(self):
        return self._new(self.data.cuda())

    def float(self, device=None):
        return self._new(self.data.float())

    def double(self, device=None):
        return self._new(self.data.double())

    def unbind(self, dim=0):
        return [self._new(x) for x in self.data.unbind(dim=dim)]


class SO3(LieGroup):
    group_name = 'SO3'
    _math = _SO3Math
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SE3):
            data = data.data[..., 3:7]
        super().__init__(data)


class SE3(LieGroup):
    group_name = 'SE3'
    _math = _SE3Math
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SO3):
            data = torch.cat([torch.zeros_like(data.data[..., :3]), data.data], dim=-1)
        super().__init__(data)

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))

    def __getitem__(self, index):
        if (_FUSED and isinstance(index, Tensor) and index.dtype == torch.int64 and index.dim() in (1, 2) and self.data.dim() == 2
                and self.data.dtype == torch.float32 and (self.data.is_cuda or _FUSED_ON_CPU) and index.device == self.data.device):
            return _GatheredSE3Proxy(self.data, index)
        return SE3(self.data[index])


class _GatheredSE3Proxy(SE3):
    """An SE3 whose rows are ``base[index]`` with ``index`` a LongTensor [P] or [P,K] -- everything an SE3 is (the gather happens
    when ``data`` is first read), plus the recognition of the skinning expression in ``act``."""

    def __init__(self, base: Tensor, index: Tensor):
        self._base, self._index, self._rows = base, index, None

    def _new(self, data: Tensor):
        return SE3(data)

    @property
    def data(self):
        if self._rows is None:
            self._rows = self._base[self._index]
        return self._rows

    @data.setter
    def data(self, v):
        self._rows = v

    def act(self, p: Tensor) -> Tensor:
        idx = self._index
        if self._rows is None and isinstance(p, Tensor) and p.dtype == torch.float32 and p.device == self._base.device and p.shape[-1] == 3:
            if idx.dim() == 2 and p.dim() == 3 and p.shape[0] == idx.shape[0] and p.shape[1] == 1:
                return _DeferredAct(self._base, idx, p[:, 0, :])                    # sk_T[indices].act(points[:, None])
            if idx.dim() == 1 and p.dim() == 2 and p.shape[0] == idx.shape[0]:
                return _se3_blend(self._base, idx[:, None], p, None)                # spT[p2sp].act(points)
        return super().act(p)


class LieGroupParameter(torch.Tensor):
    """upstream's wrapper for optimising a group element through its tangent space (not used by the reference)"""
    from torch._C import _disabled_torch_function_impl
    __torch_function__ = _disabled_torch_function_impl

    def __new__(cls, group, requires_grad=True):
        data = torch.zeros(group.tangent_shape, device=group.data.device, dtype=group.data.dtype, requires_grad=True)
        return torch.Tensor._make_subclass(cls, data, requires_grad)

    def __init__(self, group):
        self.group = group

    def retr(self):
        return self.group.retr(self)

    def log(self):
        return self.retr().log()

    def inv(self):
        return self.retr().inv()

    def adj(self, a):
        return self.retr().adj(a)

    def __mul__(self, other):
        if isinstance(other, LieGroupParameter):
            return self.retr() * other.retr()
        return self.retr() * other

    def add_(self, update, alpha):
        self.group = self.group.retr(alpha * update)

    def __getitem__(self, index):
        return self.retr().__getitem__(index)


def cat(group_list, dim):
    """concatenate groups along a batch dimension"""
    return group_list[0].__class__(torch.cat([X.data for X in group_list], dim=dim))


def stack(group_list, dim):
    return group_list[0].__class__(torch.stack([X.data for X in group_list], dim=dim))
