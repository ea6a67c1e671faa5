"""Bone-transform producer (scope row (f)-3): HIP linear / frequency-encoding kernels and the DeformMLP module against
the plain-torch restatement of SimpleDeformationNetwork (tests/test_host_cpu-style golden pinning of the restatement
itself: tests/test_oracle_golden.py::test_deform_mlp_matches_reference_modules)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_linear_kernels_match_torch():
    from sk_gs_amd import _C
    from sk_gs_amd.deform_net import _lin_bwd, _lin_fwd
    lib = _C.load_library()
    g = torch.Generator().manual_seed(0)
    for (B, in1, in2, out, relu) in [(20, 76, 0, 256, 1), (20, 256, 76, 256, 1), (20, 256, 0, 11, 0), (37, 40, 9, 50, 1),
                                     (70, 33, 0, 17, 0)]:
        X1 = torch.randn(B, in1, generator=g).cuda()
        X2 = torch.randn(B, in2, generator=g).cuda() if in2 else None
        W = (torch.randn(out, in1 + in2, generator=g) / (in1 + in2) ** 0.5).cuda().requires_grad_(True)
        b = torch.randn(out, generator=g).cuda().requires_grad_(True)
        x1r, x2r = X1.clone().requires_grad_(True), (X2.clone().requires_grad_(True) if in2 else None)
        ref = F.linear(torch.cat([x1r, x2r], dim=1) if in2 else x1r, W, b)
        ref = F.relu(ref) if relu else ref
        Y = torch.empty(B, out, device='cuda')
        _lin_fwd(lib, B, in1, in2, out, X1.data_ptr(), in1, X2.data_ptr() if in2 else None, in2, W.data_ptr(),
                 b.data_ptr(), Y.data_ptr(), out, relu)
        assert rel_err(Y, ref) <= 2e-6, (B, in1, in2, out)
        gY = torch.randn(B, out, generator=g).cuda()
        ref.backward(gY)
        gW, gb = torch.empty_like(W), torch.empty_like(b)
        gX1 = torch.empty_like(X1)
        gX2 = torch.full((B, in2), 1.0, device='cuda') if in2 else None
        _lin_bwd(lib, B, in1, in2, out, X1.data_ptr(), in1, X2.data_ptr() if in2 else None, in2, W.data_ptr(),
                 Y.data_ptr(), gY.data_ptr(), out, relu, gW.data_ptr(), gb.data_ptr(), gX1.data_ptr(), in1,
                 gX2.data_ptr() if in2 else None, in2, 2)
        assert rel_err(gW, W.grad) <= 5e-6 and rel_err(gb, b.grad) <= 5e-6 and rel_err(gX1, x1r.grad) <= 5e-6
        if in2:
            assert rel_err(gX2, x2r.grad + 1.0) <= 5e-6  # accumulated into the ones


def test_freq_encode_kernels():
    from sk_gs_amd import _C
    from sk_gs_amd.deform_net import freq_encode_torch
    lib = _C.load_library()
    g = torch.Generator().manual_seed(1)
    B, D, deg = 20, 3, 10
    x = (torch.randn(B, D, generator=g) * 0.7).cuda()
    Cdim = D + 2 * D * deg
    out = torch.zeros(B, Cdim + 5, device='cuda')
    _C._check(lib.skgs_freq_encode_forward(C.c_int32(B), C.c_int32(D), C.c_int32(deg), C.c_void_p(x.data_ptr()),
                                           C.c_int32(D), C.c_void_p(out.data_ptr()), C.c_int32(Cdim + 5), _C._stream()))
    xr = x.clone().requires_grad_(True)
    ref = freq_encode_torch(xr, deg)
    assert (out[:, :Cdim] - ref).abs().max() <= 3e-4  # sin of arguments up to 2^9 * x: argument rounding dominates
    assert float(out[:, Cdim:].abs().max()) == 0.0
    gy = torch.randn(B, Cdim + 5, generator=g).cuda()
    ref.backward(gy[:, :Cdim])
    gx = torch.empty_like(x)
    _C._check(lib.skgs_freq_encode_backward(C.c_int32(B), C.c_int32(D), C.c_int32(deg), C.c_void_p(gy.data_ptr()),
                                            C.c_void_p(out.data_ptr()), C.c_int32(Cdim + 5), C.c_void_p(gx.data_ptr()),
                                            _C._stream()))
    assert rel_err(gx, xr.grad) <= 1e-4
    # one time value broadcast to all rows (ld_x = 0)
    t = torch.tensor([0.37], device='cuda')
    out_t = torch.empty(B, 13, device='cuda')
    _C._check(lib.skgs_freq_encode_forward(C.c_int32(B), C.c_int32(1), C.c_int32(6), C.c_void_p(t.data_ptr()), C.c_int32(0),
                                           C.c_void_p(out_t.data_ptr()), C.c_int32(13), _C._stream()))
    assert (out_t - freq_encode_torch(t.view(1, 1), 6).expand(B, -1)).abs().max() <= 1e-5


def test_deform_mlp_forward_backward_matches_reference_forward():
    from sk_gs_amd.deform_net import DeformMLP
    torch.manual_seed(0)
    mlp = DeformMLP().cuda()
    joints = (torch.rand(20, 3, device='cuda') - 0.5)
    t = torch.tensor([0.41], device='cuda')
    ref = torch.cat(mlp.reference_forward(joints, t), dim=-1)
    g = torch.randn_like(ref)
    ref.backward(g)
    gref = [p.grad.clone() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    out = torch.cat(mlp(joints, t), dim=-1)
    assert rel_err(out, ref) <= 2e-5
    out.backward(g)
    for (n, p), gr in zip(mlp.named_parameters(), gref):
        assert rel_err(p.grad, gr) <= 5e-5, n
