"""Stage ``sp`` of the hot path: the SUPERPOINT LBS deform (networks/sk_gs.py:830-856).

In stage ``sp`` (30 k of the reference's 80 k default steps: ``sp_fix`` 3 000 + ``sp`` 27 000, exps/default.yaml:12-19) the
"bones" are M = 512 superpoints (exps/default.yaml:25) without a kinematic chain:

    calc_LBS_weight(points, sp_points, hyper_feature, sp_hyper_feature)   K nearest superpoints in 3 + 8 dimensions
                                                                          (sk_gs.py:751-774) + one of the four weightings
    sp_deform_net(sp_points.detach(), t)                                  DeformNetwork (sk_gs.py:209-315) on 512 rows
    d_rot = normalize(d_rotation + [0,0,0,1])                             sk_gs.py:847
    warp(..., method 'LBS'):  spT = SE3(d_xyz, d_rot);  d_points = sum_k w (spT[i_k].act(p)) - p;
                              d_rotation = sum_k w d_rot[i_k];  d_scales = sum_k w d_scaling[i_k]      (sk_gs.py:776-828)

followed by the same activation epilogue and rasterizer as stage ``sk``.  This module holds

  * ``SpDeformNet``           the reference's ``DeformNetwork`` with ``is_blender=True`` (the D-NeRF default,
                              exps/default.yaml:31): parameters under the reference's state_dict names, ``reference_forward``
                              in plain torch ops (the numerics reference of the kernels, pinned by tests/golden/
                              sp_deformnet.npz), ``forward`` on the HIP kernels of csrc/sp_mlp.hip (MFMA row-block kernels);
  * ``SuperpointGaussians``   the minimal model of the stage (parameters with the reference's names and learning-rate groups,
                              sk_gs.py:583-602) used by tests and ``bench.py --stage sp``;
  * ``FusedSuperpointStep``   the per-view training step as a straight line of C-ABI calls (the sibling of
                              ``fused_step.FusedViewStep``).

There is no CPU path: the kernels raise on host tensors; ``reference_forward`` exists for tests.
"""
import ctypes as C
import os
import math
from typing import Dict, List, Optional

import torch
from torch import Tensor, nn
import torch.nn.functional as F

from sk_gs_amd import _C
from sk_gs_amd.deform_net import freq_encode_torch


class SpDeformNet(nn.Module):
    """``DeformNetwork(D, W, is_blender=True, sep_rot=...)`` (networks/sk_gs.py:209-315).  ``sep_rot`` (the class default of the
    reference's model, `sep_rot: false` in exps/default.yaml:37): a fourth head ``local_rotation`` (:275-282,315) whose output
    ``g_rotation`` is what the skinning blends per Gaussian instead of ``d_rotation``.

    ``x_emb = freq(x, 10)`` [.,63]; ``t_emb = timenet(freq(t, 6))`` [30] with ``timenet = Linear(13,256) ReLU Linear(256,30)``
    (:250-253); ``h = [x_emb | t_emb]``; D layers ``h = relu(linear[i](h))``, after layer ``D // 2`` the input is put IN FRONT:
    ``h = [x_emb | t_emb | h]`` (:303-306); heads ``gaussian_warp`` (3), ``gaussian_scaling`` (3), ``gaussian_rotation`` (4).

    ``is_blender=False`` (the reference class's own default, :220; every shipped YAML sets true): no time network --
    ``t_emb = freq(t, t_degree)`` itself, ``1 + 2 t_degree`` columns (:255-261, :297-301 without :299); the stage then adds noise to
    the time before the call (``SuperpointGaussians.time_noise``, sk_gs.py:837-839)."""

    def __init__(self, D: int = 8, W: int = 256, p_degree: int = 10, t_degree: int = 6, time_hidden: int = 256,
                 time_out: int = 30, sep_rot: bool = False, is_blender: bool = True):
        super().__init__()
        self.D, self.W, self.p_degree, self.t_degree, self.sep_rot = D, W, p_degree, t_degree, bool(sep_rot)
        self.is_blender = bool(is_blender)
        self.skips = [D // 2]
        self.p_dim, self.t_dim = 3 * (1 + 2 * p_degree), 1 + 2 * t_degree
        self.time_hidden, self.time_out = time_hidden, time_out
        self.in_dim = self.p_dim + (time_out if self.is_blender else self.t_dim)
        if self.is_blender:
            self.timenet = nn.Sequential(nn.Linear(self.t_dim, time_hidden), nn.ReLU(inplace=True), nn.Linear(time_hidden, time_out))
        self.linear = nn.ModuleList([nn.Linear(self.in_dim, W)] + [
            nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.in_dim, W) for i in range(D - 1)])
        self.gaussian_warp, self.gaussian_scaling, self.gaussian_rotation = nn.Linear(W, 3), nn.Linear(W, 3), nn.Linear(W, 4)
        if self.sep_rot:
            self.local_rotation = nn.Linear(W, 4)
        self.reset_parameters()
        self._runners = {}

    def reset_parameters(self):
        """sk_gs.py:280-293"""
        if self.sep_rot:
            nn.init.normal_(self.local_rotation.weight, mean=0, std=1e-4)
            nn.init.zeros_(self.local_rotation.bias)
        for layer in self.linear:
            nn.init.kaiming_uniform_(layer.weight, mode='fan_in', nonlinearity='relu')
            nn.init.zeros_(layer.bias)
        nn.init.normal_(self.gaussian_warp.weight, mean=0, std=1e-5)
        nn.init.normal_(self.gaussian_scaling.weight, mean=0, std=1e-8)
        nn.init.normal_(self.gaussian_rotation.weight, mean=0, std=1e-5)
        for head in (self.gaussian_warp, self.gaussian_scaling, self.gaussian_rotation):
            nn.init.zeros_(head.bias)

    # ------------------------------------------------------------------------------------------------ plain torch
    def reference_forward(self, x: Tensor, t: Tensor) -> Dict[str, Tensor]:
        t_emb = freq_encode_torch(t.view(-1, 1), self.t_degree).expand(x.shape[0], self.t_dim)
        if self.is_blender:
            t_emb = self.timenet(t_emb)
        x_emb = freq_encode_torch(x, self.p_degree)
        h = torch.cat([x_emb, t_emb], dim=-1)
        for i, layer in enumerate(self.linear):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([x_emb, t_emb, h], -1)
        out = dict(d_xyz=self.gaussian_warp(h), d_rotation=self.gaussian_rotation(h), d_scaling=self.gaussian_scaling(h), hidden=h)
        if self.sep_rot:
            out['g_rotation'] = self.local_rotation(h)
        return out

    # ------------------------------------------------------------------------------------------------ HIP kernels
    def kernel_supported(self) -> bool:
        """csrc/sp_mlp.hip is written for the shipped configuration: 8 x 256, skip after layer 4, degrees 10 / 6, 13 -> 256 ->
        30 time network (exps/default.yaml:4-11,31)"""
        if not self.is_blender:  # raw time encoding: any degree whose columns fit the 96-column padded input
            return self.D == 8 and self.W == 256 and self.p_degree == 10 and 0 <= self.t_degree <= 15
        return (self.D == 8 and self.W == 256 and self.p_degree == 10 and self.t_degree == 6 and self.time_hidden == 256
                and self.time_out == 30)

    def runner(self, M: int, lbs_c: bool = False) -> 'SpNetRunner':
        """the launches' persistent buffers for M rows: one runner per M, kept (a network is called with 512 superpoints by the stage
        and with M * T rows by the reference's regularisers, sk_gs.py:1383-1395: replacing the runner would hand the first call's
        backward uninitialised buffers)"""
        key = M if not lbs_c else (M, 'LBS_c')
        run = self._runners.get(key)
        if run is None:
            run = self._runners[key] = SpNetRunner(self, M, lbs_c=lbs_c)
        return run

    def forward(self, x: Tensor, t: Tensor) -> Dict[str, Tensor]:
        """the reference's call ``sp_deform_net(sp_points.detach(), t)``: d_xyz, d_rotation (raw), d_scaling; autograd reaches
        the network's parameters (``x`` is detached by the caller in every stage, sk_gs.py:746-748,845)"""
        if t.numel() > 1:
            # one time PER ROW (loss_arap / loss_elastic call sp_deform_net(x [M*T,3], t [M*T,1]), sk_gs.py:1383-1395): the kernels take
            # one time for all rows -- the plain-torch body computes exactly what the reference does
            out = self.reference_forward(x, t)
            out.pop('hidden')
            return out
        params = list(self.parameters())
        out = _SpNetFn.apply(self, x, t, *params)
        res = dict(d_xyz=out[0], d_rotation=out[1], d_scaling=out[2])
        if self.sep_rot:
            res['g_rotation'] = out[3]
        return res


class _SpNetDesc(C.Structure):
    """include/skgs.h::skgs_sp_net"""
    _fields_ = [('M', C.c_int32), ('flags', C.c_int32), ('points', C.c_void_p), ('time', C.c_void_p),
                ('time_w1', C.c_void_p), ('time_b1', C.c_void_p), ('time_w2', C.c_void_p), ('time_b2', C.c_void_p),
                ('W', C.c_void_p * 8), ('b', C.c_void_p * 8),
                ('warp_w', C.c_void_p), ('warp_b', C.c_void_p), ('scaling_w', C.c_void_p), ('scaling_b', C.c_void_p),
                ('rotation_w', C.c_void_p), ('rotation_b', C.c_void_p), ('local_w', C.c_void_p), ('local_b', C.c_void_p)]


def _net_desc(net: SpDeformNet, M: int, points, time, grads: bool = False) -> _SpNetDesc:
    """pointers of the parameters (or of their .grad tensors) in the layout of include/skgs.h::skgs_sp_net"""
    def ptr(p):
        t = p.grad if grads else p
        assert t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return t.data_ptr()
    d = _SpNetDesc()
    d.M, d.flags = M, 0 if net.is_blender else (2 | (net.t_degree << 8))  # SKGS_SP_NET_RAW_TIME_DEGREE(t_degree)
    if net.sep_rot:
        d.local_w, d.local_b = ptr(net.local_rotation.weight), ptr(net.local_rotation.bias)
    d.points = None if points is None else points.data_ptr()
    d.time = None if time is None else time.data_ptr()
    if net.is_blender:
        d.time_w1, d.time_b1 = ptr(net.timenet[0].weight), ptr(net.timenet[0].bias)
        d.time_w2, d.time_b2 = ptr(net.timenet[2].weight), ptr(net.timenet[2].bias)
    for i, layer in enumerate(net.linear):
        d.W[i], d.b[i] = ptr(layer.weight), ptr(layer.bias)
    d.warp_w, d.warp_b = ptr(net.gaussian_warp.weight), ptr(net.gaussian_warp.bias)
    d.scaling_w, d.scaling_b = ptr(net.gaussian_scaling.weight), ptr(net.gaussian_scaling.bias)
    d.rotation_w, d.rotation_b = ptr(net.gaussian_rotation.weight), ptr(net.gaussian_rotation.bias)
    return d


class _SpPrepare(C.Structure):
    """include/skgs.h::skgs_sp_prepare"""
    _fields_ = [('P', C.c_int32), ('M', C.c_int32), ('K', C.c_int32), ('F', C.c_int32), ('sp_points', C.c_void_p),
                ('sp_feature', C.c_void_p), ('sp_order', C.c_void_p), ('pairs', C.c_void_p), ('pairs_bytes', C.c_size_t)]


class SpNetRunner:
    """The launches of one forward / backward of ``SpDeformNet`` on persistent buffers (csrc/sp_mlp.hip):

      forward   ONE launch, a workgroup per block of 16 superpoints runs the whole network for its rows on MFMA tiles
                (rows are independent: no exchange between workgroups), saves the activations, and ends in the stage's
                epilogue: ``bone_T = [d_xyz | normalize(d_rotation + [0,0,0,1])]`` (sk_gs.py:847), ``d_rot`` (the same unit
                quaternion, what ``warp`` blends, sk_gs.py:818-821), ``d_scale``;
      backward  launch A: the same row blocks walk the layers backwards (gZ = gY * relu', gX = gZ W) and save every gZ;
                launch B: all weight / bias gradients gW_l = gZ_l^T X_l as 64 x 64 output tiles over all 512 rows on the
                whole chip, the time network's backward in the workgroup that finishes last.
    """

    def __init__(self, net: SpDeformNet, M: int, lbs_c: bool = False):
        assert net.kernel_supported(), 'csrc/sp_mlp.hip: 8 x 256 layers, skip after layer 4, degrees 10 / 6, time net 13-256-30'
        self.net, self.M, self.lbs_c = net, int(M), bool(lbs_c)  # lbs_c: warp_method LBS_c, bone_T re-centred on the superpoints
        self.nout = 14 if net.sep_rot else 10
        self.lib = lib = _C.load_library()
        dev = next(net.parameters()).device
        if dev.type != 'cuda':
            raise _C.SkgsError('SpDeformNet kernels need the parameters on a HIP device; sk_gs_amd has no CPU path')
        lib.skgs_sp_net_saved_bytes.restype = C.c_size_t
        lib.skgs_sp_net_workspace_bytes.restype = C.c_size_t
        self.saved = torch.empty((int(lib.skgs_sp_net_saved_bytes(C.c_int32(M))),), dtype=torch.uint8, device=dev)
        self.ws = torch.zeros((int(lib.skgs_sp_net_workspace_bytes(C.c_int32(M))),), dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.bone_T, self.d_rot, self.d_scale = torch.empty((M, 7), **f32), torch.empty((M, 4), **f32), torch.empty((M, 3), **f32)
        self.raw = torch.empty((M, self.nout), **f32)  # d_xyz | d_rotation (raw) | d_scaling [| g_rotation (raw)]: the reference's outputs

    def forward(self, points: Tensor, time: Tensor, prepare: Optional[_SpPrepare] = None):
        """fills ``bone_T`` / ``d_rot`` / ``d_scale`` (and ``raw``); ``time``: 1-element device tensor.  ``prepare``: the search's
        per-step table / list preparation rides on this call's first launch (``skgs_sp_prepare``)"""
        d = _net_desc(self.net, self.M, points, time)
        d.flags |= 1 if self.lbs_c else 0  # SKGS_SP_NET_LBS_C
        self._points = points
        _C._check(self.lib.skgs_sp_net_forward(
            C.byref(d), C.c_void_p(self.raw.data_ptr()), C.c_void_p(self.bone_T.data_ptr()), C.c_void_p(self.d_rot.data_ptr()),
            C.c_void_p(self.d_scale.data_ptr()), C.c_void_p(self.saved.data_ptr()), C.c_size_t(self.saved.numel()),
            None if prepare is None else C.byref(prepare), _C._stream()))

    def backward(self, g_bone_T: Optional[Tensor], g_d_rot: Optional[Tensor], g_d_scale: Optional[Tensor],
                 g_raw: Optional[Tensor] = None, side_adam=None, saved: Optional[Tensor] = None, g_points: Optional[Tensor] = None):
        """parameter gradients WRITTEN into the parameters' ``.grad``.  Either the stage's gradients (``g_bone_T`` [M,7],
        ``g_d_rot`` [M,4], ``g_d_scale`` [M,3]: the quaternion normalisation's backward runs in the launch) or ``g_raw`` [M,10]
        w.r.t. the three raw outputs.  ``side_adam`` (``FusedAdam.side_range``): an optimizer piece for the CUs launch A
        leaves idle.  ``saved``: the activations of the forward this backward belongs to, when that is not the runner's last one (the
        autograd operator keeps a copy per call)."""
        saved = self.saved if saved is None else saved
        dg = _net_desc(self.net, self.M, None, None, grads=True)
        d = _net_desc(self.net, self.M, self._points if self.lbs_c else None, None)
        if self.lbs_c:  # the re-centring's backward needs the superpoint positions and returns their gradient [M,3] (written)
            d.flags |= 1
            dg.points = None if g_points is None else g_points.data_ptr()
        p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
        _C._check(self.lib.skgs_sp_net_backward(
            C.byref(d), C.byref(dg), p(g_bone_T), p(g_d_rot), p(g_d_scale), p(g_raw), C.c_void_p(saved.data_ptr()),
            C.c_size_t(saved.numel()), C.c_void_p(self.ws.data_ptr()), C.c_size_t(self.ws.numel()),
            None if side_adam is None else C.byref(side_adam), _C._stream()))


class _SpNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: SpDeformNet, x: Tensor, t: Tensor, *params):
        _C._require_gpu(x, 'x')
        M = x.shape[0]
        run = net.runner(M)
        x = _C._f32c(x.detach(), x.device)
        tt = _C._f32c(t.detach().reshape(-1)[:1], x.device)
        run.forward(x, tt)
        # this call's activations and its runner stay with the autograd node: a second forward before this one's backward (another
        # M, another time) overwrites the runner's buffer
        ctx.net, ctx.M, ctx.run = net, M, run
        ctx.saved_acts = run.saved.clone() if any(p.requires_grad for p in params) else None
        raw = run.raw.clone()
        if net.sep_rot:
            return raw[:, 0:3], raw[:, 3:7], raw[:, 7:10], raw[:, 10:14]
        return raw[:, 0:3], raw[:, 3:7], raw[:, 7:10]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_xyz, g_rot, g_scale, g_loc=None):
        net, M, run = ctx.net, ctx.M, ctx.run
        dev = run.raw.device
        z = lambda g, n: torch.zeros((M, n), device=dev) if g is None else g  # noqa: E731
        g_raw = torch.cat([z(g_xyz, 3), z(g_rot, 4), z(g_scale, 3)] + ([z(g_loc, 4)] if net.sep_rot else []), dim=1).contiguous()
        params = list(net.parameters())
        keep = [p.grad for p in params]
        for p in params:  # the kernels WRITE into .grad: hand autograd fresh tensors and restore what was there
            p.grad = torch.empty_like(p)
        run.backward(None, None, None, g_raw=g_raw, saved=ctx.saved_acts)
        grads = [p.grad for p in params]
        for p, k in zip(params, keep):
            p.grad = k
        return (None, None, None, *grads)


class SuperpointGaussians(nn.Module):
    """The part of ``SkeletonGaussianSplatting`` that stage ``sp`` touches per frame: the six Gaussian parameter tensors
    (gaussian_splatting.py:134-139), ``hyper_feature`` [P,8] (sk_gs.py:428,644), the superpoints ``sp_points`` [M,3] with
    ``sp_hyper_feature`` [M,8] (:458-462), the weighting's parameters (``_sp_radius`` / ``_sp_weight`` [M] or ``sp_W`` [P,M],
    :464-475) and ``sp_deform_net``.  ``forward`` reproduces the stage's call sequence on the operator path (autograd);
    ``FusedSuperpointStep`` runs the same C-ABI calls in a straight line."""

    def __init__(self, P: int, M: int = 512, K: int = 5, sh_degree: int = 3, num_frames: int = 8, seed: int = 0,
                 scale_mult: float = 1.0, lbs_method: str = 'weighted_kernel', hyper_dim: int = 8, lbs_temperature: float = 1.0,
                 warp_method: str = 'LBS', sep_rot: bool = False, is_blender: bool = True, t_degree: int = 6):
        super().__init__()
        # warp_method (sk_gs.py:776-828): 'LBS' (exps/default.yaml:36), 'LBS_c' (d_nerf_sc_gs.yaml:32: the superpoint's transform
        # re-centred on the superpoint, :803-804), 'largest' (d_nerf_sp_gs.yaml:32: each Gaussian follows the ONE superpoint with the
        # largest weight, :811-816,849-850; rotation / scale offsets stay blended).  sep_rot: the network's `g_rotation` head is what the
        # rotation blend uses (:848,818-821)
        # is_blender=False (sk_gs.py:220,255-261): the network without its time branch, and NOISE on the time of every call of the stage
        # (sk_gs.py:837-839: t + randn * time_interval * get_smooth_scale()); the caller keeps ``time_noise`` = that product current
        # (``smooth_scale`` below restates the annealing)
        assert warp_method in ('LBS', 'LBS_c', 'largest')
        self.warp_method, self.sep_rot = warp_method, bool(sep_rot)
        self.time_noise = 0.0
        from sk_gs_amd import scene
        g = scene.make_gaussians(P, seed=seed, sh_degree=sh_degree, scale_mult=scale_mult)
        self.P, self.M, self.K, self.hyper_dim = P, M, min(K, M), hyper_dim
        self.static, self.capacity = False, None
        self.max_sh_degree = self.active_sh_degree = sh_degree
        self._xyz = nn.Parameter(g['xyz'])
        self._features_dc = nn.Parameter(g['sh'][:, :1].contiguous())
        self._features_rest = nn.Parameter(g['sh'][:, 1:].contiguous())
        self._scaling = nn.Parameter(g['log_scale'])
        self._rotation = nn.Parameter(g['rot'])
        self._opacity = nn.Parameter(g['opacity_logit'])
        gen = torch.Generator().manual_seed(5000 + seed)
        assert lbs_method in ('W', 'dist', 'kernel', 'weighted_kernel')
        self.lbs_method, self.lbs_temperature = lbs_method, float(lbs_temperature)
        # superpoints: sampled from the Gaussians (init_sp_from: sampled, exps/default.yaml:33; sk_gs.py:679-701)
        pick = torch.randperm(P, generator=gen)[:M]
        # ... numbered along a Z-order curve: the numbering of the superpoints carries no meaning, and a Gaussian's neighbours
        # then share few 32-column tiles of the `W` logit table (the sparse update visits tiles, csrc/adam.hip)
        from sk_gs_amd.densify import morton_order
        pick = pick[morton_order(g['xyz'][pick])]
        self.sp_points = nn.Parameter(g['xyz'][pick].clone())
        # hyper coordinates: the reference starts them at -1e-2 / +1e-2 (sk_gs.py:644,696); a little noise keeps the 3+8-d
        # search from degenerating into the 3-d one in tests
        self.hyper_feature = nn.Parameter(torch.full((P, hyper_dim), -1e-2) + 0.02 * torch.randn(P, hyper_dim, generator=gen)) \
            if hyper_dim else None
        self.sp_hyper_feature = nn.Parameter(torch.full((M, hyper_dim), 1e-2) + 0.02 * torch.randn(M, hyper_dim, generator=gen)) \
            if hyper_dim else None
        self.sp_W = nn.Parameter(torch.randn(P, M, generator=gen)) if lbs_method == 'W' else None
        self._sp_radius = self._sp_weight = None
        if lbs_method in ('kernel', 'weighted_kernel'):  # log(0.1 * scene range + 1e-7) (sk_gs.py:698-701)
            self._sp_radius = nn.Parameter(torch.full((M,), math.log(0.1 * 2.6 + 1e-7)))
        if lbs_method == 'weighted_kernel':
            self._sp_weight = nn.Parameter(torch.zeros(M))
        torch.manual_seed(6000 + seed)
        self.sp_deform_net = SpDeformNet(sep_rot=self.sep_rot, is_blender=is_blender, t_degree=t_degree)
        with torch.no_grad():  # a trained network's output sizes instead of the 1e-5 / 1e-8 heads of reset_parameters
            if self.sep_rot:
                self.sp_deform_net.local_rotation.weight.normal_(0, 2e-3, generator=gen)
            self.sp_deform_net.gaussian_warp.weight.normal_(0, 2e-3, generator=gen)
            self.sp_deform_net.gaussian_rotation.weight.normal_(0, 2e-3, generator=gen)
            self.sp_deform_net.gaussian_scaling.weight.normal_(0, 2e-5, generator=gen)
        frames = max(num_frames, 1)
        self.register_buffer('frame_times', torch.linspace(0., 1., frames).view(frames, 1))
        self.register_buffer('_rot_bias', torch.tensor([0., 0., 0., 1.]))

    def param_groups(self, lr: float = 1e-3, spatial_scale: float = 1.0, lr_feature_scale: float = 2.5):
        """``get_params`` (gaussian_splatting.py:443-453 + sk_gs.py:583-602): the six Gaussian groups, then ``sp_deform``,
        ``sp_points``, the weighting's parameters, and the two hyper features at ``lr * lr_feature_scale``"""
        groups = [
            {'params': [self._xyz], 'lr': lr * 0.16 * spatial_scale, 'name': 'xyz'},
            {'params': [self._features_dc], 'lr': lr * 2.5, 'name': 'f_dc'},
            {'params': [self._features_rest], 'lr': lr * 2.5 / 20, 'name': 'f_rest'},
            {'params': [self._opacity], 'lr': lr * 50., 'name': 'opacity'},
            {'params': [self._scaling], 'lr': lr * 5.0, 'name': 'scaling'},
            {'params': [self._rotation], 'lr': lr * 1.0, 'name': 'rotation'},
        ]
        lr_d = lr * 0.16 * spatial_scale
        if self.hyper_feature is not None:
            groups.append({'params': [self.hyper_feature], 'lr': lr * lr_feature_scale, 'name': 'hyper'})
        if self.sp_W is not None:
            groups.append({'params': [self.sp_W], 'lr': lr_d, 'name': 'sp_W'})
        groups.append({'params': list(self.sp_deform_net.parameters()), 'lr': lr_d, 'name': 'sp_deform'})
        groups.append({'params': [self.sp_points], 'lr': lr_d, 'name': 'sp_points'})
        if self._sp_radius is not None:
            groups.append({'params': [self._sp_radius], 'lr': lr_d, 'name': 'sp_radius'})
        if self._sp_weight is not None:
            groups.append({'params': [self._sp_weight], 'lr': lr_d, 'name': 'sp_weight'})
        if self.sp_hyper_feature is not None:
            groups.append({'params': [self.sp_hyper_feature], 'lr': lr * lr_feature_scale, 'name': 'sp_hyper'})
        return groups

    # --------------------------------------------------------------------------------------------------- forward
    def superpoint_transforms(self, time_id: int, reference: bool = False):
        """``sp_deform_net(sp_points.detach(), t)`` + the normalisations of sk_gs.py:847-848 + the re-centring of `LBS_c` (:803-804):
        (spT [M,7] = (t, unit q), the rotation offsets the skinning blends [M,4] -- d_rot, or g_rot with ``sep_rot`` --, d_scale [M,3])"""
        from sk_gs_amd.skeleton import quat_act
        t = self.noisy_time(self.frame_times[time_id])
        net = self.sp_deform_net
        out = net.reference_forward(self.sp_points.detach(), t) if (reference or not self.sp_points.is_cuda) \
            else net(self.sp_points.detach(), t)
        d_rot = F.normalize(out['d_rotation'] + self._rot_bias, dim=-1)
        blend_rot = F.normalize(out['g_rotation'] + self._rot_bias, dim=-1) if self.sep_rot else d_rot
        sp_t = out['d_xyz']
        if self.warp_method == 'LBS_c':  # (the superpoints are NOT detached here: they receive a gradient, as in the reference)
            sp_t = sp_t + self.sp_points + quat_act(d_rot, -self.sp_points)
        return torch.cat([sp_t, d_rot], dim=-1), blend_rot, out['d_scaling']

    def noisy_time(self, t: Tensor) -> Tensor:
        """sk_gs.py:837-839 (and :742-744): a network without the time branch sees ``t + randn * time_interval * smooth_scale``"""
        if self.sp_deform_net.is_blender or self.time_noise == 0.0:
            return t
        return t + torch.randn_like(t) * self.time_noise

    @staticmethod
    def smooth_scale(step: int, f_s: float, annealing_steps: int, lr_final: float = 1e-15, lr_delay_steps: float = 0.01,
                     lr_delay_mult: float = 1.0) -> float:
        """``get_smooth_scale`` (sk_gs.py:723-740) for ``step`` counted from the stage's start: a LINEAR ramp from ``f_s`` to
        ``lr_final`` over ``annealing_steps`` (the reference's variable is called log_lerp; it interpolates the values themselves),
        times the sine delay; 0 when disabled.  ``time_noise = time_interval * smooth_scale(...)``"""
        if step < 0 or (f_s == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        u = min(max(step / annealing_steps, 0.0), 1.0)
        return delay * (f_s * (1 - u) + lr_final * u)

    def lbs_weights(self):
        """``calc_LBS_weight(points, sp_points, hyper_feature, sp_hyper_feature)`` (sk_gs.py:844): weights, indices"""
        from sk_gs_amd.deform import calc_lbs_weight
        return calc_lbs_weight(
            self._xyz.detach(), self.sp_points.detach(), self.K, sp_W=self.sp_W,
            kernel_radius=None if self._sp_radius is None else torch.exp(self._sp_radius),
            kernel_weight=None if self._sp_weight is None else torch.sigmoid(self._sp_weight),
            temperature=self.lbs_temperature, feature=self.hyper_feature, sp_feature=self.sp_hyper_feature)

    def forward(self, time_id: int = 0) -> Dict[str, Tensor]:
        from sk_gs_amd.deform import lbs_deform
        sh_features = torch.cat((self._features_dc, self._features_rest), dim=1)
        spT, d_rot, d_scale = self.superpoint_transforms(time_id)
        weights, indices = self.lbs_weights()
        means, scales, rotations, opacity = lbs_deform(self._xyz.detach(), weights, indices, spT, d_rot, d_scale,
                                                       self._xyz, self._scaling, self._rotation, self._opacity)
        if self.warp_method == 'largest':  # the position follows the superpoint of the largest weight alone (sk_gs.py:811-816,849-850)
            from sk_gs_amd.skeleton import se3_act
            pts = self._xyz.detach()
            self.p2sp = torch.gather(indices, -1, weights.argmax(dim=-1, keepdim=True))[:, 0]
            means = self._xyz + (se3_act(spT[self.p2sp], pts) - pts)
        return dict(points=means, opacity=opacity, scales=scales, rotations=rotations, sh_features=sh_features)

    def render(self, raster_settings, time_id: int = 0, background: Optional[Tensor] = None) -> Dict[str, Tensor]:
        from sk_gs_amd.renderer.gaussian_render import render
        out = render(**self(time_id), raster_settings=raster_settings)
        if background is not None:
            out['images'] = out['images'] + (1 - out['opacity'][None]) * background.view(3, 1, 1)
        return out

    def topology(self) -> dict:
        return {}  # (no kinematic chain in stage sp; FusedViewStep's constructor asks every model)


from sk_gs_amd.fused_step import FusedViewStep, _p  # noqa: E402  (after the model classes: fused_step imports model.py only)


class FusedSuperpointStep(FusedViewStep):
    """forward + loss + backward of one view in stage ``sp`` as a straight line of C-ABI calls on persistent buffers: the
    sibling of ``FusedViewStep`` (which covers stage ``sk``), sharing its rasterizer / loss half.

        skgs_sp_net_forward            sp_deform_net on the M superpoints + normalisation   (1 launch, MFMA row blocks)
        skgs_sp_lbs_weights_forward    K nearest superpoints in 3 + 8 dimensions + weighting (1 launch)
        skgs_lbs_deform_forward        skinning + activations                                (1 launch)
        rasterize forward, loss forward / backward, rasterize backward                       (as stage sk)
        skgs_lbs_deform_backward       -> g_weights, g_spT, g_d_rot, g_d_scale, the Gaussians' parameter gradients
        skgs_sp_lbs_weights_backward   -> hyper_feature.grad, sp_hyper_feature.grad, _sp_radius.grad, _sp_weight.grad
                                          (`W`: skgs_lbs_weights_backward -> the dense sp_W.grad)
        skgs_sp_net_backward           -> the network's parameter gradients (2 launches; the per-Gaussian rows' Adam update
                                          rides on the first one's idle CUs: ``side_optimizer``)

    Every gradient is WRITTEN into the parameter's ``.grad`` storage.  ``sp_points`` receives no gradient in this stage (the
    reference detaches it everywhere on this path, sk_gs.py:753-755,845; ``warp`` method 'LBS' does not read it): its
    ``.grad`` stays zero."""

    def __init__(self, model: SuperpointGaussians, W: int, H: int, capacity: int, lambda_dssim: float = 0.2,
                 background: Optional[Tensor] = None, grad_scale: float = 1.0, densify_stats: bool = False,
                 tile_bucket: int = 0, view_table=None):
        assert isinstance(model.sp_deform_net, SpDeformNet) and model.sp_deform_net.kernel_supported()
        model.sk_deform_net = None  # (what FusedViewStep's constructor probes for the stage-sk network)
        super().__init__(model, W, H, capacity, lambda_dssim=lambda_dssim, background=background, grad_scale=grad_scale,
                         densify_stats=densify_stats, tile_bucket=tile_bucket, view_table=None)
        lib, dev = self.lib, model._xyz.device
        P, M, K = self.P, self.M, self.K
        self.F = int(model.hyper_dim) if model.hyper_feature is not None else 0
        assert self.F in (0, 8), 'csrc/sp_knn.hip: 0 or 8 hyper dimensions'
        # warp_method: LBS_c is the network launches' business (bone_T re-centred in their epilogue), `largest` the skinning's (the
        # position follows the bone of the largest weight: skgs_deform_inputs.largest)
        self.largest = model.warp_method == 'largest'
        self.time_noise = torch.full((1,), float(model.time_noise), dtype=torch.float32, device=dev)
        self.net = model.sp_deform_net.runner(M, lbs_c=model.warp_method == 'LBS_c')
        self.nn_dist = torch.empty((P, K), dtype=torch.float32, device=dev)
        lib.skgs_sp_lbs_weights_workspace_bytes.restype = C.c_size_t
        self.spw_ws = torch.empty((max(int(lib.skgs_sp_lbs_weights_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(self.F))),
                                       16),), dtype=torch.uint8, device=dev)
        # inverse neighbour lists (filed by the search, walked by the backward) and the backward's payload / partial buffers
        for fn in (lib.skgs_sp_pairs_bytes, lib.skgs_sp_skinning_backward_workspace_bytes):
            fn.restype = C.c_size_t
        self.pairs = torch.zeros((int(lib.skgs_sp_pairs_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(K))),), dtype=torch.uint8, device=dev)
        self.sb_ws = torch.empty((int(lib.skgs_sp_skinning_backward_workspace_bytes(C.c_int32(P), C.c_int32(M), C.c_int32(K))),),
                                 dtype=torch.uint8, device=dev)
        self.g_d_rot = torch.empty((M, 4), dtype=torch.float32, device=dev)
        self.g_d_scale = torch.empty((M, 3), dtype=torch.float32, device=dev)
        self.view_table = view_table
        if view_table is not None:
            vs = view_table.settings
            assert (vs.image_height, vs.image_width) == (self.H, self.W)
        self.wide = True
        # LBS_method 'W': with ``sparse_logits`` (FusedSuperpointTrainStep) the dense [P, M] logit gradient is never written -- the
        # table's Adam update visits only the 32-column tiles a row has ever been touched in (skgs_adam_logit_rows)
        self.sparse_logits = False
        # cotangents from OUTSIDE the step on the LBS weights [P,K] / the superpoint transforms [M,7] (the reference's loss reads
        # outputs['_knn_w'] and outputs['_spT']: sk_gs_amd/reference_fused.py): added inside the rows pass / before the network's backward
        self.g_weights_extra = self.g_bone_T_extra = None
        self.logit_mask = torch.zeros((P,), dtype=torch.int32, device=dev) if model.sp_W is not None else None
        self.sp_order = torch.empty((M,), dtype=torch.int32, device=dev)
        self.sp_rank = torch.empty((M,), dtype=torch.int32, device=dev)
        self.indices.zero_()  # (the search reads the previous call's nearest superpoint as its starting hint)
        self.refresh_scan_order()

    @torch.no_grad()
    def refresh_scan_order(self):
        """the order the search scans the superpoints in: along a Z-order curve of their positions (any order gives the same
        result; this one lets a wave skip most candidates together).  In place -- a captured step keeps the pointer; call it
        when the superpoints have moved far (e.g. with the densification cadence)."""
        from sk_gs_amd.densify import morton_order
        order = morton_order(self.model.sp_points.detach())
        self.sp_order.copy_(order.to(torch.int32))
        self.sp_rank[order] = torch.arange(order.numel(), dtype=torch.int32, device=order.device)  # the inverse permutation

    # ---- the pieces FusedViewStep asks its subclass for --------------------------------------------------------------
    def table_grad_span(self):
        return None

    def _zero_table_grads(self):
        pass  # no per-frame tables in stage sp

    def _deform_inputs(self, time_id):
        m = self.model
        a = _C._DeformInputs()
        a.P, a.K, a.M = self.P, self.K, self.M
        a.points = a.xyz = m._xyz.data_ptr()  # points = xyz.detach() (sk_gs.py:833)
        a.weights, a.indices = self.weights.data_ptr(), self.indices.data_ptr()
        a.bone_T, a.bone_drot, a.bone_dscale = self.net.bone_T.data_ptr(), self.net.d_rot.data_ptr(), self.net.d_scale.data_ptr()
        a.log_scale, a.rot, a.opacity_logit = m._scaling.data_ptr(), m._rotation.data_ptr(), m._opacity.data_ptr()
        a.live_count = None
        a.largest = 1 if self.largest else 0
        return a

    def _time(self, time_id) -> Tensor:
        if time_id is not None:
            t = self.model.frame_times[time_id]
        else:
            from sk_gs_amd import view_slot as vsl
            t = self.view_table.slot[vsl.W_TIME:vsl.W_TIME + 1]
        if not self.model.sp_deform_net.is_blender:
            # the stage's time noise (sk_gs.py:837-839), drawn on the device so that a captured step draws a fresh one per replay
            # (torch's generator is graph-safe); its scale is a device scalar the loop updates with set_time_noise()
            t = (t.reshape(1) + torch.randn(1, device=t.device) * self.time_noise).contiguous()
        return t

    def set_time_noise(self, scale: float):
        """``time_interval * get_smooth_scale()`` of the current iteration (is_blender=False only; a 4-byte fill, legal between replays)"""
        self.time_noise.fill_(float(scale))

    @torch.no_grad()
    def forward(self, rs=None, time_id=None):
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        P, M, K = self.P, self.M, self.K
        assert (rs is None) == (time_id is None) and (rs is not None or self.view_table is not None)
        # (the search's table and list counters are prepared by extra workgroups of the network's first launch)
        prep = _SpPrepare(P, M, K, self.F, m.sp_points.data_ptr(), None if m.sp_hyper_feature is None else m.sp_hyper_feature.data_ptr(),
                          self.sp_order.data_ptr(), self.pairs.data_ptr(), self.pairs.numel())
        if os.environ.get('SKGS_SP_SEPARATE_PREPARE'):  # (A/B measurements: the search prepares its table in a launch of its own)
            prep = None
        self.net.forward(m.sp_points, self._time(time_id), prepare=prep)
        chk(lib.skgs_sp_lbs_weights_forward(
            C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_int32(self.F), _p(m._xyz), _p(m.hyper_feature), _p(m.sp_points),
            _p(m.sp_hyper_feature), _p(m._sp_radius), _p(m._sp_weight), C.c_float(m.lbs_temperature), _p(m.sp_W),
            _p(self.sp_order), _p(self.sp_rank), _p(self.indices), _p(self.weights), _p(self.nn_dist), _p(self.pairs), C.c_size_t(self.pairs.numel()),
            C.c_int32(0), C.c_int32(0 if prep is None else 1), st))
        d = self._deform_inputs(time_id)
        a = self._raster_inputs(rs)
        if self.deform_in_preprocess:  # the skinning as a job of the rasterizer's per-Gaussian launch (joints = NULL: no search)
            j = _C._KnnDeformJob()
            j.M, j.K, j.largest = M, K, d.largest
            j.points, j.bone_T, j.bone_drot, j.bone_dscale = d.points, d.bone_T, d.bone_drot, d.bone_dscale
            j.xyz, j.log_scale, j.rot, j.opacity_logit = d.xyz, d.log_scale, d.rot, d.opacity_logit
            j.out_idx, j.out_weights = self.indices.data_ptr(), self.weights.data_ptr()
            j.means, j.scales = self.means.data_ptr(), self.scales.data_ptr()
            j.rotations, j.opacity = self.rotations.data_ptr(), self.opacity.data_ptr()
            a.deform_job = C.pointer(j)
        else:
            chk(lib.skgs_lbs_deform_forward(C.byref(d), _p(self.means), _p(self.scales), _p(self.rotations), _p(self.opacity),
                                            None, None, None, st))
        chk(lib.skgs_rasterize_forward(C.byref(a), C.byref(self._bufs), _p(self.radii), _p(self.image),
                                       _p(self.out_opacity), None, None, st))
        a.deform_job = None  # (the backward's copy of the inputs: the job was the forward's)
        return a, d

    def _attach_backward_job(self, g, d, time_id):
        """the ROWS pass of the skinning + weighting backward as a job of the rasterizer's per-Gaussian backward launch
        (skgs_raster_grads.sp_skinning_job: the arguments of skgs_sp_skinning_backward below without the upstream gradients);
        the bones and finalize launches follow inside skgs_rasterize_backward.  SKGS_SEPARATE_DEFORM_BACKWARD=1: off"""
        if not self.deform_backward_in_preprocess:
            return None
        m = self.model
        gp = lambda t: None if t is None or t.grad is None else t.grad.data_ptr()  # noqa: E731
        pp = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        logits = m.sp_W is not None
        j = _C._SpSkinningJob()
        j.in_, j.F = C.pointer(d), self.F
        j.feature, j.sp_feature, j.sp_radius_raw, j.sp_weight_raw = pp(m.hyper_feature), pp(m.sp_hyper_feature), pp(m._sp_radius), pp(m._sp_weight)
        j.temperature, j.logit_weighting, j.nn_dist = float(m.lbs_temperature), 1 if logits else 0, self.nn_dist.data_ptr()
        j.g_weights = self.g_weights.data_ptr() if logits else None
        j.g_xyz, j.g_log_scale, j.g_rot = m._xyz.grad.data_ptr(), m._scaling.grad.data_ptr(), m._rotation.grad.data_ptr()
        j.g_opacity_logit, j.g_feature = m._opacity.grad.data_ptr(), None if logits else gp(m.hyper_feature)
        j.g_bone_T, j.g_bone_drot, j.g_bone_dscale = self.g_bone_T.data_ptr(), self.g_d_rot.data_ptr(), self.g_d_scale.data_ptr()
        j.g_sp_feature, j.g_sp_radius, j.g_sp_weight = None if logits else gp(m.sp_hyper_feature), gp(m._sp_radius), gp(m._sp_weight)
        j.pairs, j.pairs_bytes = self.pairs.data_ptr(), self.pairs.numel()
        j.workspace, j.workspace_bytes = self.sb_ws.data_ptr(), self.sb_ws.numel()
        j.g_weights_extra = None if self.g_weights_extra is None else self.g_weights_extra.data_ptr()
        g.sp_skinning_job = C.cast(C.pointer(j), C.c_void_p)
        self._rows_backward_done = True
        return j

    @torch.no_grad()
    def backward_skinning(self, time_id=None, part=None):
        assert part is None
        lib, m, st, chk = self.lib, self.model, _C._stream(), _C._check
        P, M, K = self.P, self.M, self.K
        d = self._deform_inputs(time_id)
        g = lambda t: None if t is None else _p(t.grad)  # noqa: E731
        logits = m.sp_W is not None
        # skinning + weighting backward: rows | bones (the inverse lists of the forward) | finalize -- no atomics
        if self._rows_backward_done:  # (ran with the rasterizer's backward: _attach_backward_job; the flag stays until the next
            pass                       # backward_raster clears it -- see FusedViewStep.backward_skinning)
        else:
            assert self.g_weights_extra is None, 'a cotangent on the weights rides on the rows pass of the rasterizer backward (sp_skinning_job)'
            chk(lib.skgs_sp_skinning_backward(
                C.byref(d), C.c_int32(self.F), _p(m.hyper_feature), _p(m.sp_hyper_feature), _p(m._sp_radius), _p(m._sp_weight),
                C.c_float(m.lbs_temperature), C.c_int32(1 if logits else 0), _p(self.nn_dist), _p(self.g_means), _p(self.g_scales),
                _p(self.g_rotations), _p(self.g_opacity), _p(self.g_weights) if logits else None, _p(m._xyz.grad), _p(m._scaling.grad),
                _p(m._rotation.grad), _p(m._opacity.grad), None if logits else g(m.hyper_feature), _p(self.g_bone_T), _p(self.g_d_rot),
                _p(self.g_d_scale), None if logits else g(m.sp_hyper_feature), g(m._sp_radius), g(m._sp_weight), _p(self.pairs),
                C.c_size_t(self.pairs.numel()), _p(self.sb_ws), C.c_size_t(self.sb_ws.numel()), st))
        if logits and not self.sparse_logits:  # `W`: the dense [P,M] logit gradient (what autograd's gather backward builds)
            chk(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(M), C.c_int32(K), _p(self.weights), _p(self.indices),
                                              _p(self.g_weights), _p(m.sp_W.grad), st))
        if self.g_bone_T_extra is not None:
            self.g_bone_T.add_(self.g_bone_T_extra)
        side = None
        if self.side_optimizer is not None:  # the per-Gaussian rows' Adam update on the CUs the row-block launch leaves idle
            side = self.side_optimizer[0].side_range(*self.side_optimizer[1:])
        # (LBS_c: the re-centring's backward writes d loss / d sp_points; nothing else on this path reaches the superpoints' positions)
        self.net.backward(self.g_bone_T, self.g_d_rot, self.g_d_scale, side_adam=side,
                          g_points=self.model.sp_points.grad if self.net.lbs_c else None)

    @torch.no_grad()
    def refresh_logit_mask(self, optimizer):
        """``sparse_logits``: rebuild the live-tile mask of the logit table from its Adam moments -- after a restored state, a
        re-ordering of the Gaussians or any other change of the table's rows from outside the step"""
        m = self.model
        st = optimizer.state[m.sp_W]
        _C._check(self.lib.skgs_adam_logit_mask_rebuild(C.c_int32(self.P), C.c_int32(self.M), _p(st['exp_avg']), _p(st['exp_avg_sq']),
                                                        _p(self.logit_mask), _C._stream()))

    @torch.no_grad()
    def sparse_logit_adam(self, optimizer):
        """the logit table's piece of the Adam step (``optimizer.step(['sp_W'], advance=False)``), from this step's neighbours,
        weights and their cotangent; ``sp_W.grad`` is neither read nor written"""
        m = self.model
        _C._check(self.lib.skgs_adam_logit_rows(
            C.c_int32(self.P), C.c_int32(self.M), C.c_int32(self.K), _p(self.weights), _p(self.indices), _p(self.g_weights),
            C.c_void_p(optimizer.table_entry(m.sp_W)), _p(self.logit_mask), C.c_double(optimizer.betas[0]),
            C.c_double(optimizer.betas[1]), C.c_double(optimizer.eps), C.c_void_p(optimizer.step_state.data_ptr()), C.c_int32(0),
            _C._stream()))

    def status(self) -> dict:
        """(synchronising) the rasterizer's status words + ``pairs_overflow``: a superpoint's inverse list outgrew its capacity in the
        LAST step (its pairs beyond the capacity were dropped from the backward's sums), and ``pairs_overflow_events``: the number of
        steps in which that happened since this object was built (sticky: a loop that asks every N steps still sees it)"""
        st = _C.read_status(self.geom)
        words = self.pairs[:12].view(torch.int32).tolist()
        st['pairs_overflow'], st['pairs_overflow_events'] = words[1], words[2]
        return st


class FusedSuperpointTrainStep:
    """One training step of one rank in stage ``sp``: ``FusedSuperpointStep.forward_backward`` + ``FusedAdam.step`` with the
    per-Gaussian rows' update (xyz, SH, opacity, scaling, rotation, hyper features: > 95 % of the optimizer's bytes without
    a dense logit table) as the side job of the network's two backward launches, and one closing launch for the rest (network,
    superpoint tables) that advances the counter and, with an ordered ``ViewTable``, selects the next view.  Same arithmetic
    as ``step.forward_backward(); optimizer.step()``."""

    ROW_GROUPS = ('xyz', 'f_dc', 'f_rest', 'opacity', 'scaling', 'rotation', 'hyper')
    # LBS_method 'W' (exps/default.yaml:35): the dense [P, M] logit table is 7.5 x the other rows together (1.4 GB of optimizer
    # traffic per step at P = 100k, M = 512; beside the network's launches it streamed at 4.4 TB/s on the CUs they leave, as a
    # launch of its own at the Adam kernel's 5.9 TB/s).  Its update is its own piece of the step: sparse (the tiles a row has
    # ever been touched in, no dense gradient: csrc/adam.hip::adam_logit_rows_kernel) where the table fits the kernel, dense
    # otherwise (``sparse_logits=False``)
    WIDE_GROUPS = ('sp_W',)

    def __init__(self, step: FusedSuperpointStep, optimizer, enable: bool = True, sparse_logits: bool = True):
        self.step, self.optimizer = step, optimizer
        names = [g.get('name') for g in optimizer.param_groups]
        self.rows = [n for n in names if n in self.ROW_GROUPS]
        self.wide = [n for n in names if n in self.WIDE_GROUPS]
        self.rest = [n for n in names if n not in self.ROW_GROUPS and n not in self.WIDE_GROUPS]
        self.fused = bool(enable and self.rows and self.rest and len(optimizer._chunk_ranges(self.rows)) == 1
                          and len(optimizer._chunk_ranges(self.rest)) == 1)
        step.side_optimizer = (optimizer, self.rows, None) if self.fused else None
        step.sparse_logits = bool(self.fused and self.wide and sparse_logits and step.M <= 1024 and step.K <= 16)
        if step.sparse_logits:
            step.refresh_logit_mask(optimizer)
            # ... and again whenever the table's moments change from outside a step (a restored checkpoint, optimizer surgery): the
            # sparse update is bit-identical to the dense one only while the mask covers every tile with a non-zero moment
            import weakref
            ref_step, ref_opt = weakref.ref(step), weakref.ref(optimizer)

            def _refresh():
                st, op = ref_step(), ref_opt()
                if st is None or op is None or not st.sparse_logits:
                    return False
                if st.model.sp_W in op.state and st.P == st.model.sp_W.shape[0]:
                    st.refresh_logit_mask(op)
                return True
            optimizer.add_state_listener(_refresh)

    def __call__(self, rs=None, time_id=None, target=None):
        self.step.forward_backward(rs, time_id, target)
        if self.fused:
            vt = self.step.view_table
            if self.step.sparse_logits:
                self.step.sparse_logit_adam(self.optimizer)
            elif self.wide:
                self.optimizer.step(self.wide, advance=False)
            self.optimizer.step_tail(self.rest, next_view=vt.advance() if (vt is not None and rs is None) else None)
        else:
            self.optimizer.step()
