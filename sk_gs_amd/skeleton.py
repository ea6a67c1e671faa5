"""Bone-transform producer in plain torch (M <= 512 rows: latency only).

Restates, on [.., 7] = (t, q_xyzw) tensors, what the reference does with lietorch objects:
  SO3 / SE3 product and action          my_ext/_C/include/lie.h:54-64,242-246
  kinematic()                           networks/sk_gs.py:1069-1107   (T_i(p) = R_i (p - j_i) + j_i)
  skeleton_warp_SE3()                   networks/sk_gs.py:193-206     (pointer jumping over the ancestor table)
  joint_parents table                   my_ext/_C/src/nerf/sp_gs_joint.cu:55-85 (2^l-th ancestor, filled with root)
All functions are differentiable torch code; autograd supplies the backward.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor
import torch.nn.functional as F


def quat_mul(a: Tensor, b: Tensor) -> Tensor:
    """Hamilton product, xyzw order (Eigen quaternion product used by lie.h:54-56)"""
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def quat_act(q: Tensor, p: Tensor) -> Tensor:
    """rotate p by the (unit) quaternion q: p + w*uv + v x uv, uv = 2 v x p (lie.h:59-64)"""
    v, w = q[..., :3], q[..., 3:]
    uv = 2 * torch.linalg.cross(v, p)
    return p + w * uv + torch.linalg.cross(v, uv)


def se3_mul(a: Tensor, b: Tensor) -> Tensor:
    """(R_a R_b, t_a + R_a t_b) with re-normalised rotation (lie.h:45-47,242-244)"""
    qa = F.normalize(a[..., 3:], dim=-1)
    qb = F.normalize(b[..., 3:], dim=-1)
    q = F.normalize(quat_mul(qa, qb), dim=-1)
    t = a[..., :3] + quat_act(qa, b[..., :3])
    return torch.cat([t, q], dim=-1)


def se3_act(T: Tensor, p: Tensor) -> Tensor:
    return quat_act(F.normalize(T[..., 3:], dim=-1), p) + T[..., :3]


def axis_angle_to_quat(r: Tensor) -> Tensor:
    """SO3.exp of an axis-angle vector -> xyzw quaternion"""
    theta = r.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    half = 0.5 * theta
    return torch.cat([r / theta * torch.sin(half), torch.cos(half)], dim=-1)


def build_ancestor_table(parents: Tensor, root: int) -> Tuple[Tensor, int]:
    """[M, L] table whose column l holds the 2^l-th ancestor (saturating at the root), L = ceil(log2(max depth))."""
    M = parents.shape[0]
    par = parents.clone().long()
    par[root] = root
    depth = torch.zeros(M, dtype=torch.long)
    for i in range(M):
        d, f = 0, i
        while f != root:
            f = int(par[f])
            d += 1
        depth[i] = d
    max_depth = int(depth.max()) if M > 0 else 0
    L = 0
    while (1 << L) < max_depth:
        L += 1
    L = max(L, 1)
    table = torch.full((M, L), root, dtype=torch.long)
    table[:, 0] = par
    for l in range(1, L):
        table[:, l] = table[table[:, l - 1], l - 1]
    return table, max_depth


def root_constants(M: int, root: int, device=None) -> Tuple[Tensor, Tensor]:
    """(identity 7-vector, [M,1] bool mask of the root): built once, outside any graph capture (H2D copies)"""
    ident = torch.tensor([0, 0, 0, 0, 0, 0, 1.], device=device)
    mask = torch.zeros(M, 1, dtype=torch.bool)
    mask[root] = True
    return ident, mask.to(device)


def skeleton_warp_se3(local_T: Tensor, global_T: Optional[Tensor], ancestors: Tensor, root: int,
                      consts: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """global bone transforms T_i = G * prod_{a in path(root -> i)} T_a; the root's own transform is forced to identity"""
    M, L = ancestors.shape
    ident, mask = consts if consts is not None else root_constants(M, root, local_T.device)
    out = torch.where(mask, ident.expand(M, 7), local_T)
    for level in range(L):
        out = se3_mul(out[ancestors[:, level]], out)
    # after ceil(log2(depth)) doublings every node has absorbed 2^L >= depth ancestors (root = identity pads)
    if global_T is None:
        return out
    return se3_mul(global_T.view(1, 7).expand(M, 7), out)


def kinematic(joints: Tensor, sk_r: Tensor, g_tr: Optional[Tensor], ancestors: Tensor, root: int,
              consts: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """joint rotations (unit xyzw quaternions [M,4]) about their joint positions -> global SE3 [M,7]
    (sk_gs.py:1090-1106): sk_t = joints + R(-joints)"""
    sk_t = joints + quat_act(sk_r, -joints)
    return skeleton_warp_se3(torch.cat([sk_t, sk_r], dim=-1), g_tr, ancestors, root, consts)


# ------------------------------------------------------------------------------------------------ fused HIP path
def build_topology(parents: Tensor, root: int, device=None) -> dict:
    """Skeleton topology in the form the ``bone_chain`` kernels walk: direct parents, bones sorted by depth and the
    start of every depth level (level 0 = {root}).  Built on the host once per skeleton (the reference rebuilds its
    ancestor table in ``joint_discovery`` every 1000 iterations, sk_gs.py:1245-1265)."""
    M = parents.shape[0]
    par = parents.clone().long().cpu()
    par[root] = root
    depth = torch.zeros(M, dtype=torch.long)
    for i in range(M):
        d, f = 0, i
        while f != root:
            f = int(par[f])
            d += 1
            assert d <= M, 'parents do not form a tree rooted at `root`'
        depth[i] = d
    order = torch.argsort(depth, stable=True)
    num_levels = int(depth.max()) + 1 if M > 0 else 1
    counts = torch.bincount(depth, minlength=num_levels)
    level_start = torch.zeros(num_levels + 1, dtype=torch.long)
    level_start[1:] = torch.cumsum(counts, 0)
    i32 = dict(dtype=torch.int32, device=device)
    return dict(parents=par.to(**i32), level_nodes=order.to(**i32), level_start=level_start.to(**i32), root=int(root),
                num_levels=num_levels)


class _BoneChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sk_r_raw, joints, global_T, topo):
        from sk_gs_amd import _C
        bone_T, chain = _C.bone_chain_forward(sk_r_raw, joints, global_T, topo, save_chain=True)
        ctx.topo = topo
        ctx.has_global = global_T is not None
        ctx.save_for_backward(sk_r_raw, joints, global_T, chain)
        return bone_T

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_bone_T):
        from sk_gs_amd import _C
        sk_r_raw, joints, global_T, chain = ctx.saved_tensors
        need_j = ctx.needs_input_grad[1]
        g_raw, g_j, g_g = _C.bone_chain_backward(sk_r_raw, joints, global_T, ctx.topo, chain, g_bone_T, need_joints=need_j)
        return g_raw, g_j, g_g, None


def bone_chain(sk_r_raw: Tensor, joints: Tensor, global_T: Optional[Tensor], topo: dict) -> Tensor:
    """Fused ``normalize(raw + [0,0,0,1]) -> kinematic -> skeleton_warp_SE3`` on the GPU (csrc/bone_chain.hip):
    global bone transforms [M,7] from the raw joint-rotation outputs; one kernel per direction instead of ~450."""
    return _BoneChain.apply(sk_r_raw, joints, global_T, topo)
