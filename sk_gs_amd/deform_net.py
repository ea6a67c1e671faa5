"""Bone-transform producer of the skeleton stage (scope row (f)-3).

``DeformMLP`` mirrors ``SimpleDeformationNetwork`` (networks/sk_gs.py:134-164): frequency encoding of the joint
positions (degree 10) and of the time (degree 6), ``MLP_with_skips`` (my_ext/blocks/mlp.py:43-85: 8 x 256, ReLU, the
input concatenated again after layer 4) and three linear heads (4 | 4 | 3 = raw joint rotation, d_rot, d_scale,
``sk_dims``).  The parameter names match the reference's ``state_dict`` (``dynamic_net.net.{i}.weight``,
``dynamic_net.last.{j}.weight`` ...), the three heads are stored as one [11, 256] matrix so that a forward or backward
pass is one launch per layer (csrc/mlp.hip) instead of ~65 torch launches for 20 rows.

``forward`` runs the HIP kernels (device tensors only, no fallback); ``reference_forward`` is the same network in
plain torch ops -- the numerics reference of the kernels and the CPU restatement used by the host tests.

Two kernel paths, same results up to summation order: ``FusedDeformMLP`` -- the whole network as ONE persistent launch per
direction (csrc/mlp_fused.hip; <= 48 rows, the skeleton stage's one row per bone), the default wherever it applies -- and
``DeformMLPRunner`` -- one launch per layer (csrc/mlp.hip; any row count, e.g. the 512 superpoints of the sp stage).
"""
import ctypes as C
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn
import torch.nn.functional as F

from sk_gs_amd import _C


def freq_encode_torch(x: Tensor, degree: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(deg-1) x), cos(2^(deg-1) x)] (freqencoder.cu:7-33: cos as sin(. + pi/2))"""
    outs = [x]
    for f in range(degree):
        outs.append(torch.sin(x * (2.0 ** f)))
        outs.append(torch.sin(x * (2.0 ** f) + torch.pi / 2))
    return torch.cat(outs, dim=-1)


class _MLPWithSkips(nn.Module):
    """parameter container with the reference's names: ``net`` (hidden layers) and ``last`` (heads, stored fused)"""

    def __init__(self, in_channels: int, dim_hidden: int, out_channels: Sequence[int], num_layers: int, skips: Sequence[int]):
        super().__init__()
        self.in_channels, self.dim_hidden, self.num_layers = in_channels, dim_hidden, num_layers
        self.out_channels, self.skips = tuple(out_channels), tuple(skips)
        net, c = [], in_channels
        for i in range(num_layers):
            net.append(nn.Linear(c, dim_hidden))
            c = dim_hidden + (in_channels if i in self.skips else 0)
        self.net = nn.ModuleList(net)
        self.head_in = c
        heads = [nn.Linear(c, oc) for oc in self.out_channels]  # same initialisation as three separate nn.Linear
        self.last_weight = nn.Parameter(torch.cat([h.weight.data for h in heads], dim=0))
        self.last_bias = nn.Parameter(torch.cat([h.bias.data for h in heads], dim=0))

    # ---- the reference stores the heads as ``last.{j}.weight`` / ``last.{j}.bias``
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        w, b = destination.pop(prefix + 'last_weight'), destination.pop(prefix + 'last_bias')
        o = 0
        for j, oc in enumerate(self.out_channels):
            destination[f'{prefix}last.{j}.weight'], destination[f'{prefix}last.{j}.bias'] = w[o:o + oc], b[o:o + oc]
            o += oc

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        keys = [f'{prefix}last.{j}.weight' for j in range(len(self.out_channels))]
        if all(k in state_dict for k in keys):
            state_dict[prefix + 'last_weight'] = torch.cat([state_dict.pop(k) for k in keys], dim=0)
            state_dict[prefix + 'last_bias'] = torch.cat(
                [state_dict.pop(f'{prefix}last.{j}.bias') for j in range(len(self.out_channels))], dim=0)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def layer_dims(self) -> List[Tuple[int, int]]:
        """(in1, in2) per hidden layer: in2 = width of the re-injected input"""
        dims, skip_in = [], 0
        for i in range(self.num_layers):
            dims.append((self.in_channels if i == 0 else self.dim_hidden, skip_in))
            skip_in = self.in_channels if i in self.skips else 0
        return dims + [(self.dim_hidden, skip_in)]  # last entry: the heads


class DeformMLP(nn.Module):
    def __init__(self, p_in_channels: int = 3, t_in_channels: int = 1, out_channels: Sequence[int] = (4, 4, 3),
                 width: int = 256, depth: int = 8, skips: Sequence[int] = (4,), p_degree: int = 10, t_degree: int = 6):
        super().__init__()
        self.p_in, self.t_in, self.p_degree, self.t_degree = p_in_channels, t_in_channels, p_degree, t_degree
        self.p_dim = p_in_channels * (1 + 2 * p_degree)
        self.t_dim = t_in_channels * (1 + 2 * t_degree)
        self.dynamic_net = _MLPWithSkips(self.p_dim + self.t_dim, width, out_channels, depth, skips)

    # ------------------------------------------------------------------------------------------------ plain torch
    def reference_forward(self, points: Tensor, t: Tensor) -> List[Tensor]:
        net = self.dynamic_net
        p_embed = freq_encode_torch(points, self.p_degree)
        t_embed = freq_encode_torch(t.view(-1, self.t_in), self.t_degree).expand(points.shape[0], -1)
        x0 = torch.cat([p_embed, t_embed], dim=-1)
        x = x0
        for i in range(net.num_layers):
            x = F.relu(net.net[i](x))
            if i in net.skips:
                x = torch.cat([x, x0], dim=-1)
        out = F.linear(x, net.last_weight, net.last_bias)
        return list(out.split(net.out_channels, dim=-1))

    # ------------------------------------------------------------------------------------------------ HIP kernels
    def forward(self, points: Tensor, t: Tensor) -> List[Tensor]:
        net = self.dynamic_net
        params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
        out = _DeformMLPFn.apply(self, torch.is_grad_enabled(), points, t, *params)
        return list(out.split(net.out_channels, dim=-1))


def _lin_fwd(lib, B, in1, in2, out, X1, ldx1, X2, ldx2, W, b, Y, ldy, relu):
    _C._check(lib.skgs_linear_forward(C.c_int32(B), C.c_int32(in1), C.c_int32(in2), C.c_int32(out), C.c_void_p(X1),
                                      C.c_int32(ldx1), C.c_void_p(X2), C.c_int32(ldx2), C.c_void_p(W), C.c_void_p(b),
                                      C.c_void_p(Y), C.c_int32(ldy), C.c_int32(relu), _C._stream()))


def _lin_bwd(lib, B, in1, in2, out, X1, ldx1, X2, ldx2, W, Y, gY, ldy, relu, gW, gb, gX1, ldg1, gX2, ldg2, acc2):
    _C._check(lib.skgs_linear_backward(C.c_int32(B), C.c_int32(in1), C.c_int32(in2), C.c_int32(out), C.c_void_p(X1),
                                       C.c_int32(ldx1), C.c_void_p(X2), C.c_int32(ldx2), C.c_void_p(W), C.c_void_p(Y),
                                       C.c_void_p(gY), C.c_int32(ldy), C.c_int32(relu), C.c_void_p(gW), C.c_void_p(gb),
                                       C.c_void_p(gX1), C.c_int32(ldg1), C.c_void_p(gX2), C.c_int32(ldg2),
                                       C.c_int32(acc2), _C._stream()))


class DeformMLPRunner:
    """The launch sequence of one forward / backward on caller-provided buffers (shared by the autograd Function and by
    ``FusedViewStep``).  ``x0`` [B, IN], ``acts`` [L, B, H], ``out`` [B, OUT]; gradients are WRITTEN to the given tensors."""

    def __init__(self, mlp: DeformMLP):
        self.mlp, self.lib = mlp, _C.load_library()

    def encode(self, points: Tensor, t: Tensor, x0: Tensor):
        m, lib = self.mlp, self.lib
        B, ld = points.shape[0], x0.shape[1]
        _C._check(lib.skgs_freq_encode_forward(C.c_int32(B), C.c_int32(m.p_in), C.c_int32(m.p_degree),
                                               C.c_void_p(points.data_ptr()), C.c_int32(m.p_in),
                                               C.c_void_p(x0.data_ptr()), C.c_int32(ld), _C._stream()))
        # the time embedding is the same for every row (ld_x = 0)
        _C._check(lib.skgs_freq_encode_forward(C.c_int32(B), C.c_int32(m.t_in), C.c_int32(m.t_degree),
                                               C.c_void_p(t.data_ptr()), C.c_int32(0),
                                               C.c_void_p(x0[:, m.p_dim:].data_ptr()), C.c_int32(ld), _C._stream()))

    def forward_hidden(self, x0: Tensor, acts: Tensor):
        net, lib = self.mlp.dynamic_net, self.lib
        B, IN, H = x0.shape[0], net.in_channels, net.dim_hidden
        dims = net.layer_dims()
        prev, ldp = x0.data_ptr(), IN
        for i, layer in enumerate(net.net):
            in1, in2 = dims[i]
            _lin_fwd(lib, B, in1, in2, H, prev, ldp, x0.data_ptr() if in2 else None, IN, layer.weight.data_ptr(),
                     layer.bias.data_ptr(), acts[i].data_ptr(), H, 1)
            prev, ldp = acts[i].data_ptr(), H

    def forward(self, x0: Tensor, acts: Tensor, out: Tensor):
        net, lib = self.mlp.dynamic_net, self.lib
        self.forward_hidden(x0, acts)
        in1, in2 = net.layer_dims()[-1]
        _lin_fwd(lib, x0.shape[0], in1, in2, out.shape[1], acts[-1].data_ptr(), net.dim_hidden,
                 x0.data_ptr() if in2 else None, net.in_channels, net.last_weight.data_ptr(), net.last_bias.data_ptr(),
                 out.data_ptr(), out.shape[1], 0)

    def backward_hidden(self, x0: Tensor, acts: Tensor, grads: List[Tensor], g_act: Tensor, g_x0: Optional[Tensor] = None,
                        g_x0_started: bool = False):
        """hidden layers, last to first; ``g_act[0]`` holds dL/d(last activation) on entry.  ``grads``: [gW0, gb0, ...]
        (written).  ``g_x0`` [B, IN]: if given, receives dL/d(encoded input) -- the first layer's and every skip layer's
        contribution (joints trained through the network's input, networks/sk_gs.py:604-607,1073); ``g_x0_started``: the
        heads already wrote into it."""
        net, lib = self.mlp.dynamic_net, self.lib
        B, IN, H, L = x0.shape[0], net.in_channels, net.dim_hidden, net.num_layers
        dims = net.layer_dims()
        x0p = x0.data_ptr()
        gx0p = None if g_x0 is None else g_x0.data_ptr()
        started = g_x0_started
        cur = 0
        for i in range(L - 1, -1, -1):
            in1, in2 = dims[i]
            layer = net.net[i]
            x1, ld1 = (acts[i - 1].data_ptr(), H) if i > 0 else (x0p, IN)
            if i > 0:
                gx1, ldg1, gx2 = g_act[1 - cur].data_ptr(), H, (gx0p if in2 else None)
                acc = 2 if (gx2 is not None and started) else 0
                started = started or gx2 is not None
            else:  # the first layer's input IS the encoded input
                gx1, ldg1, gx2 = gx0p, IN, None
                acc = 1 if (gx1 is not None and started) else 0
            _lin_bwd(lib, B, in1, in2, H, x1, ld1, x0p if in2 else None, IN, layer.weight.data_ptr(), acts[i].data_ptr(),
                     g_act[cur].data_ptr(), H, 1, grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr(), gx1, ldg1, gx2, IN,
                     acc)
            cur = 1 - cur

    def backward(self, x0: Tensor, acts: Tensor, out: Tensor, g_out: Tensor, grads: List[Tensor], g_act: Tensor,
                 g_x0: Optional[Tensor] = None):
        """``grads``: [gW0, gb0, ..., gW_last, gb_last] (written); ``g_act`` [2, B, H] ping-pong scratch; ``g_x0`` [B, IN]
        (written) or None"""
        net, lib = self.mlp.dynamic_net, self.lib
        B, IN, H, L = x0.shape[0], net.in_channels, net.dim_hidden, net.num_layers
        in1, in2 = net.layer_dims()[-1]
        gx2 = g_x0.data_ptr() if (g_x0 is not None and in2) else None
        _lin_bwd(lib, B, in1, in2, out.shape[1], acts[L - 1].data_ptr(), H, x0.data_ptr() if in2 else None, IN,
                 net.last_weight.data_ptr(), None, g_out.data_ptr(), out.shape[1], 0, grads[-2].data_ptr(),
                 grads[-1].data_ptr(), g_act[0].data_ptr(), H, gx2, IN, 0)
        self.backward_hidden(x0, acts, grads[:-2], g_act, g_x0, g_x0_started=gx2 is not None)

    def input_grad(self, g_x0: Tensor, x0: Tensor, g_points: Tensor, accumulate: bool = False):
        """dL/d(points) [B, p_in] from dL/d(encoded input): the frequency-encoding backward (freqencoder.cu:36-60) over the
        point part of the encoding (the time part belongs to no parameter)"""
        m = self.mlp
        _C._check(self.lib.skgs_freq_encode_backward(
            C.c_int32(x0.shape[0]), C.c_int32(m.p_in), C.c_int32(m.p_degree), C.c_void_p(g_x0.data_ptr()),
            C.c_void_p(x0.data_ptr()), C.c_int32(x0.shape[1]), C.c_void_p(g_points.data_ptr()), C.c_int32(int(accumulate)),
            _C._stream()))


class _MlpLayer(C.Structure):
    _fields_ = [('W', C.c_void_p), ('bias', C.c_void_p), ('gW', C.c_void_p), ('gb', C.c_void_p),
                ('in_hidden', C.c_int32), ('in_x0', C.c_int32), ('out', C.c_int32), ('relu', C.c_int32)]


MLP_MAX_LAYERS = 12


class _MlpDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('p_dim', C.c_int32), ('p_degree', C.c_int32), ('t_dim', C.c_int32),
                ('t_degree', C.c_int32), ('hidden', C.c_int32), ('n_layers', C.c_int32),
                ('layer', _MlpLayer * MLP_MAX_LAYERS), ('n_heads', C.c_int32), ('head_dim', C.c_int32 * 4),
                ('head_out', C.c_void_p * 4), ('head_gout', C.c_void_p * 4)]


class BoneChainDesc(C.Structure):
    """include/skgs.h::skgs_bone_chain_desc: the kinematic chain riding on the fused network launches"""
    _fields_ = [('M', C.c_int32), ('root', C.c_int32), ('num_levels', C.c_int32),
                ('parents', C.c_void_p), ('level_nodes', C.c_void_p), ('level_start', C.c_void_p),
                ('joints', C.c_void_p), ('global_T', C.c_void_p), ('frame_index', C.c_void_p),
                ('bone_T', C.c_void_p), ('chain_A', C.c_void_p), ('sk_r_raw', C.c_void_p), ('g_bone_T', C.c_void_p),
                ('g_joints', C.c_void_p), ('g_global_T', C.c_void_p), ('sk_cache', C.c_void_p)]


def fused_supported(mlp: 'DeformMLP', B: int) -> bool:
    """the one-launch kernels are built for the skeleton stage's shapes: <= 48 rows, hidden width 256, encoded input a
    multiple of 4 and <= 128 wide, <= 10 layers with the heads (skgs.h: skgs_deform_mlp_forward)"""
    net = mlp.dynamic_net
    return (1 <= B <= 48 and net.dim_hidden == 256 and net.in_channels <= 128 and net.in_channels % 4 == 0
            and net.num_layers + 1 <= 10 and sum(net.out_channels) <= net.dim_hidden)


class FusedDeformMLP:
    """``SimpleDeformationNetwork.forward`` / its backward as one persistent launch each (csrc/mlp_fused.hip) on persistent
    buffers: ``x0`` [B, IN], ``acts`` [L, B, H], ``out`` [B, OUT]; the backward WRITES the weight gradients into the
    tensors given to ``backward`` (normally the parameters' ``.grad``) and optionally dL/dx0."""

    def __init__(self, mlp: DeformMLP, B: int):
        if not fused_supported(mlp, B):
            raise _C.SkgsError(f'fused deform MLP: unsupported shape (rows {B}, hidden {mlp.dynamic_net.dim_hidden}, '
                               f'input {mlp.dynamic_net.in_channels}); use DeformMLPRunner')
        self.mlp, self.lib, self.B = mlp, _C.load_library(), int(B)
        net = mlp.dynamic_net
        dev = net.last_weight.device
        if dev.type != 'cuda':
            raise _C.SkgsError('fused deform MLP needs the network on a HIP device; sk_gs_amd has no CPU path')
        f32 = dict(dtype=torch.float32, device=dev)
        self.x0 = torch.empty((B, net.in_channels), **f32)
        self.acts = torch.empty((net.num_layers, B, net.dim_hidden), **f32)
        self.out = torch.empty((B, sum(net.out_channels)), **f32)
        self.lib.skgs_deform_mlp_workspace_bytes.restype = C.c_size_t
        d = self._desc(None)
        nbytes = self.lib.skgs_deform_mlp_workspace_bytes(C.byref(d))
        if nbytes == 0:
            _C._check(1)
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=dev)  # exchange images + launch counters
        _C._check(self.lib.skgs_deform_mlp_workspace_init(C.c_void_p(self.workspace.data_ptr()), C.c_size_t(nbytes),
                                                          _C._stream()))

    def _desc(self, grads: Optional[Sequence[Tensor]], head_out: Optional[Sequence[Tensor]] = None,
              head_gout: Optional[Sequence[Tensor]] = None) -> _MlpDesc:
        """the launch descriptor.  Its static part (shapes, weight and bias addresses) is built once and kept for as long as
        the parameters stay where they are (filling ~100 ctypes fields per call was a third of the operator path's skeleton
        stage); gradient and head addresses are patched in per call"""
        m, net = self.mlp, self.mlp.dynamic_net
        flat = getattr(self, '_flat_params', None)
        if flat is None:
            flat = self._flat_params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
        key = tuple(p.data_ptr() for p in flat)
        d = getattr(self, '_desc_cache', None)
        if d is None or self._desc_key != key:
            d = _MlpDesc()
            d.B, d.p_dim, d.p_degree, d.t_dim, d.t_degree = self.B, m.p_in, m.p_degree, m.t_in, m.t_degree
            d.hidden, d.n_layers = net.dim_hidden, net.num_layers + 1
            dims = net.layer_dims()
            for i, (in1, in2) in enumerate(dims):
                w, b = flat[2 * i], flat[2 * i + 1]
                assert w.is_contiguous() and b.is_contiguous() and w.dtype == torch.float32
                L = d.layer[i]
                L.W, L.bias = w.data_ptr(), b.data_ptr()
                L.in_hidden, L.in_x0 = (0, in1) if i == 0 else (in1, in2)
                L.out, L.relu = w.shape[0], int(i < net.num_layers)
            self._desc_cache, self._desc_key = d, key
        if grads is not None:
            for i in range(d.n_layers):
                gw, gb, w, b = grads[2 * i], grads[2 * i + 1], flat[2 * i], flat[2 * i + 1]
                assert gw.is_contiguous() and gb.is_contiguous() and gw.shape == w.shape and gb.shape == b.shape
                d.layer[i].gW, d.layer[i].gb = gw.data_ptr(), gb.data_ptr()
        heads = head_out if head_out is not None else head_gout
        d.n_heads = 0
        if heads is not None:  # the heads as separate [B, dim] tensors (sk_r | d_rot | d_scale)
            assert len(heads) == len(net.out_channels) <= 4
            d.n_heads = len(heads)
            for j, (h, oc) in enumerate(zip(heads, net.out_channels)):
                assert h.is_cuda and h.is_contiguous() and h.dtype == torch.float32 and tuple(h.shape) == (self.B, oc)
                d.head_dim[j] = oc
                (d.head_out if head_out is not None else d.head_gout)[j] = h.data_ptr()
                (d.head_gout if head_out is not None else d.head_out)[j] = None  # (nothing of an earlier call stays behind)
        return d

    def forward(self, points: Tensor, t: Tensor, head_out: Optional[Sequence[Tensor]] = None,
                out: Optional[Tensor] = None, bones: Optional[BoneChainDesc] = None, side_adam=None) -> Tensor:
        """points [B, p_in], t: device tensor with t_in floats -> ``out`` (default ``self.out``) [B, OUT], or the heads
        written to the separate tensors ``head_out`` (also fills x0 / acts).  ``bones`` (needs ``head_out``): the kinematic
        chain runs in the same launch (``skgs_skeleton_forward``: rows = bones, head 0 = raw joint rotations) and fills
        ``bones.bone_T`` / ``bones.chain_A``; ``side_adam`` (``FusedAdam.side_range``): an optimizer piece for the CUs that
        launch leaves idle."""
        assert points.is_cuda and points.is_contiguous() and points.dtype == torch.float32 and points.shape[0] == self.B
        assert t.is_cuda and t.dtype == torch.float32 and t.numel() == self.mlp.t_in
        out = self.out if out is None else out
        assert out.is_cuda and out.is_contiguous() and out.shape == self.out.shape and out.dtype == torch.float32
        d = self._desc(None, head_out=head_out)
        if bones is not None:
            assert head_out is not None
            _C._check(self.lib.skgs_skeleton_forward(
                C.byref(d), C.byref(bones), C.c_void_p(points.data_ptr()), C.c_void_p(t.data_ptr()),
                C.c_void_p(self.x0.data_ptr()), C.c_void_p(self.acts.data_ptr()), C.c_void_p(self.workspace.data_ptr()),
                C.c_size_t(self.workspace.numel()), None if side_adam is None else C.byref(side_adam), _C._stream()))
            return out
        assert side_adam is None
        _C._check(self.lib.skgs_deform_mlp_forward(
            C.byref(d), C.c_void_p(points.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(self.x0.data_ptr()),
            C.c_void_p(self.acts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(self.workspace.data_ptr()),
            C.c_size_t(self.workspace.numel()), _C._stream()))
        return out

    def backward(self, points: Tensor, t: Tensor, g_out, grads: Sequence[Tensor], g_x0: Optional[Tensor] = None,
                 reencode: bool = False, side_adam=None, bones: Optional[BoneChainDesc] = None):
        """``g_out``: [B, OUT] or one tensor per head; ``grads``: [gW0, gb0, ..., gW_heads, gb_heads] (written); ``g_x0``
        [B, IN] (written) or None.  The encoded input is the copy the last ``forward`` left in ``self.x0`` (same points and
        time!); ``reencode``: let the kernel rebuild it from ``points`` / ``t`` instead.  ``side_adam``
        (``optim.AdamRange``): an optimizer piece that runs on the CUs this launch leaves idle -- parameters whose
        gradients are final and that the network does not touch.  ``bones`` (with ``g_out`` = the heads' gradient
        tensors): the kinematic chain's backward runs inside the launch (``skgs_skeleton_backward``) -- the gradient of head
        0 (raw joint rotations) is then an OUTPUT (``g_out[0]`` receives a copy), computed from ``bones.g_bone_T``."""
        if isinstance(g_out, Tensor):
            assert g_out.is_cuda and g_out.is_contiguous() and g_out.shape == self.out.shape
            d = self._desc(grads)
        else:
            d, g_out = self._desc(grads, head_gout=g_out), None
        if bones is not None:
            assert g_out is None, 'skeleton backward: pass the heads\' gradients as separate tensors'
            _C._check(self.lib.skgs_skeleton_backward(
                C.byref(d), C.byref(bones), C.c_void_p(points.data_ptr()), C.c_void_p(t.data_ptr()),
                C.c_void_p(None if reencode else self.x0.data_ptr()), C.c_void_p(self.acts.data_ptr()),
                C.c_void_p(None if g_x0 is None else g_x0.data_ptr()), C.c_void_p(self.workspace.data_ptr()),
                C.c_size_t(self.workspace.numel()), None if side_adam is None else C.byref(side_adam), _C._stream()))
            return
        _C._check(self.lib.skgs_deform_mlp_backward_adam(
            C.byref(d), C.c_void_p(points.data_ptr()), C.c_void_p(t.data_ptr()),
            C.c_void_p(None if reencode else self.x0.data_ptr()), C.c_void_p(self.acts.data_ptr()),
            C.c_void_p(None if g_out is None else g_out.data_ptr()), C.c_void_p(None if g_x0 is None else g_x0.data_ptr()),
            C.c_void_p(self.workspace.data_ptr()), C.c_size_t(self.workspace.numel()),
            None if side_adam is None else C.byref(side_adam), _C._stream()))

    def status(self) -> dict:
        """(synchronising) forward / backward launches so far and 'failed': launches whose in-kernel exchange timed out
        (must be 0); ``one_xcd_*``: launches whose in-launch census found all of the network's workgroups on one XCD and kept the
        exchange inside its L2 (``skgs_deform_mlp_xcd_mode``); ``xcds_*``: the XCDs workgroup 0 of a launch has run on so far"""
        w = self.workspace[:32].view(torch.int32).cpu()
        return dict(forward=int(w[0]), backward=int(w[3]), failed=int(w[1]), one_xcd_forward=int(w[4]), one_xcd_backward=int(w[5]),
                    xcds_forward=[x for x in range(8) if int(w[6]) >> x & 1], xcds_backward=[x for x in range(8) if int(w[7]) >> x & 1])


_FUSED_POOLS = weakref.WeakKeyDictionary()  # DeformMLP -> {(rows, device, stream): [free FusedDeformMLP runners]}


class _DeformMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp: DeformMLP, track: bool, points: Tensor, t: Tensor, *params):
        _C._require_gpu(points, 'points')
        net = mlp.dynamic_net
        ctx.need_points = points.requires_grad  # joints trained through the network input (sk_gs.py:604-607)
        points = points.detach().float().contiguous()
        t = t.detach().float().reshape(-1).contiguous().to(points.device)
        B = points.shape[0]
        f32 = dict(dtype=torch.float32, device=points.device)
        if fused_supported(mlp, B) and not getattr(mlp, 'force_layered', False):
            # a runner (exchange workspace, saved activations) is busy from a forward until its backward has run: free
            # ones are kept on the module, per row count and stream, instead of being allocated and initialised per call
            key = (B, points.device.index, torch.cuda.current_stream().cuda_stream)
            pool = _FUSED_POOLS.setdefault(mlp, {}).setdefault(key, [])
            run = pool.pop() if pool else FusedDeformMLP(mlp, B)
            ctx.pool = pool
            # the returned tensor must not be owned by anything the context references: ctx -> run -> out -> grad_fn -> ctx
            # would be a reference cycle that keeps the whole autograd graph (and its AccumulateGrad nodes, bound to the
            # stream of THIS call) alive until the next gc -- which breaks a later hipGraph capture of the step
            out = run.forward(points, t, out=torch.empty_like(run.out))
            if not (track and any(ctx.needs_input_grad)):  # inference (no_grad): no backward will hand the runner back
                ctx.fused = None
                if len(pool) < 4:
                    pool.append(run)
                return out
            ctx.mlp, ctx.fused, ctx.pt = mlp, run, (points, t)
            return out
        ctx.fused = None
        x0 = torch.empty((B, net.in_channels), **f32)
        acts = torch.empty((net.num_layers, B, net.dim_hidden), **f32)
        out = torch.empty((B, sum(net.out_channels)), **f32)
        run = DeformMLPRunner(mlp)
        run.encode(points, t, x0)
        run.forward(x0, acts, out)
        ctx.mlp = mlp
        ctx.save_for_backward(x0, acts, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        mlp = ctx.mlp
        net = mlp.dynamic_net
        params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
        grads = [torch.empty_like(p) for p in params]
        g_points = None
        if ctx.fused is not None:
            run = ctx.fused
            g_x0 = torch.empty_like(run.x0) if ctx.need_points else None
            run.backward(ctx.pt[0], ctx.pt[1], g_out.contiguous(), grads, g_x0)
            if g_x0 is not None:
                g_points = torch.empty_like(ctx.pt[0])
                DeformMLPRunner(mlp).input_grad(g_x0, run.x0, g_points)
            ctx.fused = None
            if len(ctx.pool) < 4:
                ctx.pool.append(run)  # (same stream: the next forward's launches are ordered behind this backward's)
            return (None, None, g_points, None) + tuple(grads)
        x0, acts, out = ctx.saved_tensors
        g_act = torch.empty((2,) + tuple(acts.shape[1:]), dtype=torch.float32, device=acts.device)
        g_x0 = torch.empty_like(x0) if ctx.need_points else None
        run = DeformMLPRunner(mlp)
        run.backward(x0, acts, out, g_out.contiguous(), grads, g_act, g_x0)
        if g_x0 is not None:
            g_points = torch.empty((x0.shape[0], mlp.p_in), dtype=torch.float32, device=x0.device)
            run.input_grad(g_x0, x0, g_points)
        return (None, None, g_points, None) + tuple(grads)  # the time is an input, not a parameter


class _SkeletonStageFn(torch.autograd.Function):
    """The whole skeleton stage of a frame on the operator path -- deform network over the joints, ``normalize(raw + [0, 0, 0,
    1])``, the kinematic chain (sk_gs.py:1069-1107) and the frame's row of the test-time cache (:1077-1079) -- as ONE launch
    per direction (``skgs_skeleton_forward`` / ``skgs_skeleton_backward``: what ``FusedViewStep`` runs), instead of
    ``_DeformMLPFn`` + three copies of its strided heads + ``skeleton._BoneChain`` + seven small torch launches for the cache
    row + a ``select`` of the global transform and its backward.  Returns (bone_T [M,7], d_rot [M,4], d_scale [M,3])."""

    @staticmethod
    def forward(ctx, mlp: DeformMLP, track: bool, joints: Tensor, t: Tensor, global_tr: Optional[Tensor], time_id: int,
                topo: dict, cache_row: Optional[Tensor], *params):
        _C._require_gpu(joints, 'joints')
        net = mlp.dynamic_net
        M = joints.shape[0]
        ctx.need_points = joints.requires_grad
        pts = joints.detach().float().contiguous()
        tt = t.detach().float().reshape(-1).contiguous().to(pts.device)
        f32 = dict(dtype=torch.float32, device=pts.device)
        key = (M, pts.device.index, torch.cuda.current_stream().cuda_stream)
        pool = _FUSED_POOLS.setdefault(mlp, {}).setdefault(key, [])
        run = pool.pop() if pool else FusedDeformMLP(mlp, M)
        heads = [torch.empty((M, oc), **f32) for oc in net.out_channels]
        bone_T, chain_A = torch.empty((M, 7), **f32), torch.empty((M, 7), **f32)
        b = _SkeletonStageFn._desc(pts, global_tr, time_id, topo)
        b.bone_T, b.chain_A = bone_T.data_ptr(), chain_A.data_ptr()
        b.sk_cache = cache_row.data_ptr() if cache_row is not None else None
        run.forward(pts, tt, head_out=heads, bones=b)
        if not (track and any(ctx.needs_input_grad)):  # inference (no_grad): no backward will hand the runner back
            if len(pool) < 4:
                pool.append(run)
            return bone_T, heads[1], heads[2]
        ctx.mlp, ctx.run, ctx.pool, ctx.topo, ctx.time_id = mlp, run, pool, topo, time_id
        ctx.pt = (pts, tt, global_tr.detach() if global_tr is not None else None, heads[0], chain_A)
        return bone_T, heads[1], heads[2]

    @staticmethod
    def _desc(pts: Tensor, global_tr: Optional[Tensor], time_id: int, topo: dict) -> BoneChainDesc:
        b = BoneChainDesc()
        b.M, b.root, b.num_levels = pts.shape[0], topo['root'], topo['num_levels']
        b.parents, b.level_nodes, b.level_start = (topo['parents'].data_ptr(), topo['level_nodes'].data_ptr(),
                                                   topo['level_start'].data_ptr())
        b.joints = pts.data_ptr()
        b.global_T = global_tr[time_id].data_ptr() if global_tr is not None else None
        b.frame_index = None
        return b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_bone_T, g_d_rot, g_d_scale):
        mlp, run = ctx.mlp, ctx.run
        net = mlp.dynamic_net
        pts, tt, global_tr, sk_r_raw, chain_A = ctx.pt
        M = pts.shape[0]
        f32 = dict(dtype=torch.float32, device=pts.device)
        params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
        grads = [torch.empty_like(p) for p in params]

        def dense(g, oc):
            return torch.zeros((M, oc), **f32) if g is None else g.float().contiguous()

        g_heads = [torch.empty((M, net.out_channels[0]), **f32), dense(g_d_rot, net.out_channels[1]),
                   dense(g_d_scale, net.out_channels[2])]  # (head 0's gradient is produced by the chain's backward)
        g_x0 = torch.empty_like(run.x0) if ctx.need_points else None
        g_joints = torch.empty((M, 3), **f32) if ctx.need_points else None
        g_global = torch.zeros_like(global_tr) if (global_tr is not None and ctx.needs_input_grad[4]) else None
        b = _SkeletonStageFn._desc(pts, global_tr, ctx.time_id, ctx.topo)
        b.chain_A, b.sk_r_raw = chain_A.data_ptr(), sk_r_raw.data_ptr()
        g_bT = dense(g_bone_T, 7)
        b.g_bone_T = g_bT.data_ptr()
        b.g_joints = g_joints.data_ptr() if g_joints is not None else None
        b.g_global_T = g_global[ctx.time_id].data_ptr() if g_global is not None else None
        run.backward(pts, tt, g_heads, grads, g_x0, bones=b)
        if g_x0 is not None:  # joints also feed the network: + the frequency encoding's backward
            DeformMLPRunner(mlp).input_grad(g_x0, run.x0, g_joints, accumulate=True)
        ctx.run = None
        if len(ctx.pool) < 4:
            ctx.pool.append(run)
        return (None, None, g_joints, None, g_global, None, None, None) + tuple(grads)


def skeleton_stage(mlp: DeformMLP, joints: Tensor, t: Tensor, global_tr: Optional[Tensor], time_id: int, topo: dict,
                   cache_row: Optional[Tensor] = None):
    """(bone_T [M,7], d_rot [M,4], d_scale [M,3]) of a frame: deform network + kinematic chain in one launch per direction
    (``_SkeletonStageFn``).  ``global_tr``: the [frames, 7] table (row ``time_id`` is used; its gradient comes back as a
    table with that row filled) or None; ``cache_row``: where the launch writes [normalised joint rotation | d_rot |
    d_scale] (the reference's ``sk_cache[time_id]``), or None."""
    net = mlp.dynamic_net
    params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
    return _SkeletonStageFn.apply(mlp, torch.is_grad_enabled(), joints, t, global_tr, time_id, topo, cache_row, *params)


def skeleton_stage_supported(mlp: 'DeformMLP', M: int) -> bool:
    net = mlp.dynamic_net
    return fused_supported(mlp, M) and tuple(net.out_channels) == (4, 4, 3) and not getattr(mlp, 'force_layered', False)
