"""Optional accelerators for an UNMODIFIED checkout of the reference, applied AFTER it has been imported::

    import sk_gs_amd
    sk_gs_amd.install_reference_hooks()          # before `import networks`: the compiled ops and the two third-party stand-ins
    import networks, train
    sk_gs_amd.accelerate_reference()             # after: two methods of the reference's classes get a fast path

``install_reference_hooks`` makes the reference RUN on an MI355X; with it alone two pieces of its training step are still long chains
of small torch launches.  ``accelerate_reference`` replaces exactly those two methods -- same arguments, same returned objects, and a
call that does not match the fast path's conditions is handed to the reference's own method:

* ``networks.losses.ssim.SSIM_Loss.forward`` (ssim.py:26-43: five depth-wise 11x11 convolutions + ~30 element-wise kernels, 3.5 ms per
  800x800 image on this GPU, forward + backward) -> the fused kernels of ``sk_gs_amd.losses.image_loss`` (``skgs_image_loss_forward /
  _backward``: 41 us) for one fp32 3-channel image pair on a HIP device, window 11, reduction 'mean';
* ``networks.sk_gs.SkeletonGaussianSplatting.kinematic`` (sk_gs.py:1069-1107: ``sk_deform_net`` -> ``SO3`` -> ``sk_t = joints +
  sk_r.act(-joints)`` -> ``skeleton_warp_SE3``, ~60 Lie-group launches forward and backward through the lietorch stand-in) -> the
  reference's own network call followed by ONE launch per direction, ``skgs_bone_chain_forward / _backward``
  (``sk_gs_amd.skeleton.bone_chain``), for the training-time quaternion path (``which_rotation: quaternion``, no ``sk_r_delta``, no
  ``sk_feature``, a 7-vector or no global transform).  The result is handed back as ``SE3.InitFromVec(T)`` of the stand-in, so
  everything the reference does with ``sk_T`` afterwards -- ``sk_T[indices].act(...)``, ``sk_T.vec()`` -- is unchanged; the train-time
  cache write ``self.sk_cache[time_id] = ...`` (:1077-1079) is kept.

The reference's files are not touched; ``restore_reference()`` puts the original methods back.
"""
from __future__ import annotations

import sys

import torch
import torch.nn.functional as F

_originals = {}
calls = {'ssim_fused': 0, 'ssim_reference': 0, 'kinematic_fused': 0, 'kinematic_reference': 0}  # counters (tests)


# ------------------------------------------------------------------------------------------------ SSIM_Loss.forward
def _as_chw(img: torch.Tensor):
    """one image as [3,H,W] contiguous, or None when it is not one fp32 3-channel image on a HIP device"""
    if not (img.is_cuda and img.dtype == torch.float32):
        return None
    if img.dim() == 4 and img.shape[0] == 1:
        img = img[0]
    if img.dim() != 3:
        return None
    if img.shape[-1] == 3 and img.shape[0] != 3:      # HWC, as sk_gs.loss hands it over (sk_gs.py:1527-1529)
        return img.permute(2, 0, 1).contiguous()
    if img.shape[0] == 3:
        return img.contiguous()
    return None


def ssim_loss_forward(self, img1, img2):
    """``SSIM_Loss.forward`` (networks/losses/ssim.py:26-43): ``1 - mean SSIM`` of one image pair through the fused kernels"""
    if getattr(self, 'window_size', 11) == 11 and getattr(self, 'reduction', 'mean') == 'mean' and img1.shape == img2.shape:
        a, b = _as_chw(img1), _as_chw(img2)
        if a is not None and b is not None:
            from sk_gs_amd.losses import image_loss
            calls['ssim_fused'] += 1
            return image_loss(a, b, 0.0, 1.0)     # lambda_l1 * L1 + lambda_ssim * (1 - SSIM) with (0, 1)
    calls['ssim_reference'] += 1
    return _originals['ssim'](self, img1, img2)


# ------------------------------------------------------------------------------------------------ SkeletonGaussianSplatting.kinematic
_topo_cache = {}


def _topology(parents_table: torch.Tensor, root) -> dict:
    """``build_topology`` of the skeleton in ``joint_parents`` (column 0: the direct parent, sp_gs_joint.cu:55-85) / ``joint_root``,
    cached until the table is rewritten (joint discovery runs every 1000+ iterations)"""
    from sk_gs_amd.skeleton import build_topology
    key = (parents_table.data_ptr(), parents_table._version, tuple(parents_table.shape), str(parents_table.device))
    hit = _topo_cache.get(key)
    if hit is None:
        _topo_cache.clear()
        r = int(root.reshape(-1)[0]) if torch.is_tensor(root) else int(root)
        hit = _topo_cache[key] = build_topology(parents_table[:, 0].long().cpu(), r, parents_table.device)
    return hit


def kinematic(self, joints, t, g_tr=None, time_id=None, sk_r_delta=None):
    """``SkeletonGaussianSplatting.kinematic`` (networks/sk_gs.py:1069-1107) with the Lie-group chain as one launch per direction"""
    lie = sys.modules.get('lietorch')
    fast = (joints.is_cuda and joints.dtype == torch.float32 and sk_r_delta is None and getattr(self, 'sk_feature', None) is None
            and (self.training or not self.test_time_interpolate) and getattr(self, '_R_dim', 4) == 4 and lie is not None
            and hasattr(lie, 'fused_calls')                                   # this package's stand-in, not upstream lietorch
            and (g_tr is None or (torch.is_tensor(g_tr) and g_tr.dim() == 1 and g_tr.shape[0] == 7))
            and self.joint_parents.dim() == 2 and joints.shape[0] <= 512)
    if not fast:
        calls['kinematic_reference'] += 1
        return _originals['kinematic'](self, joints, t, g_tr, time_id, sk_r_delta)
    from sk_gs_amd.skeleton import bone_chain
    sk_r_raw, d_rot, d_scale = self.sk_deform_net(joints, t)                  # the reference's own network call (:1074)
    if sk_r_raw.shape[-1] != 4:
        calls['kinematic_reference'] += 1
        return _originals['kinematic'](self, joints, t, g_tr, time_id, sk_r_delta)
    if self.training and time_id is not None:                                  # the train-time cache (:1077-1079)
        with torch.no_grad():
            sk_r = F.normalize(sk_r_raw + sk_r_raw.new_tensor([0., 0., 0., 1.]), dim=-1)
            self.sk_cache[time_id] = torch.cat([sk_r, d_rot, d_scale], dim=-1)
    topo = _topology(self.joint_parents, self.joint_root)
    T = bone_chain(sk_r_raw, joints, g_tr, topo)                               # [M,7] = (t, q_xyzw): kinematic + skeleton_warp_SE3
    calls['kinematic_fused'] += 1
    return lie.SE3.InitFromVec(T), d_rot, d_scale


# ------------------------------------------------------------------------------------------------ install / restore
def accelerate_reference(ssim: bool = True, kinematic_chain: bool = True) -> list:
    """Patch the two methods on the reference's classes (the modules must be imported already).  Returns what was patched."""
    done = []
    if ssim:
        mod = sys.modules.get('networks.losses.ssim')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.losses.ssim is not loaded)")
        if 'ssim' not in _originals:
            _originals['ssim'] = mod.SSIM_Loss.forward
            mod.SSIM_Loss.forward = ssim_loss_forward
        done.append('networks.losses.ssim.SSIM_Loss.forward')
    if kinematic_chain:
        mod = sys.modules.get('networks.sk_gs')
        if mod is None:
            raise RuntimeError("accelerate_reference(): import the reference first (networks.sk_gs is not loaded)")
        if 'kinematic' not in _originals:
            _originals['kinematic'] = mod.SkeletonGaussianSplatting.kinematic
            mod.SkeletonGaussianSplatting.kinematic = kinematic
        done.append('networks.sk_gs.SkeletonGaussianSplatting.kinematic')
    return done


def restore_reference():
    """put the reference's own methods back"""
    if 'ssim' in _originals and 'networks.losses.ssim' in sys.modules:
        sys.modules['networks.losses.ssim'].SSIM_Loss.forward = _originals.pop('ssim')
    if 'kinematic' in _originals and 'networks.sk_gs' in sys.modules:
        sys.modules['networks.sk_gs'].SkeletonGaussianSplatting.kinematic = _originals.pop('kinematic')
