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
                                            C.c_int32(0), _C._stream()))
    assert rel_err(gx, xr.grad) <= 1e-4
    _C._check(lib.skgs_freq_encode_backward(C.c_int32(B), C.c_int32(D), C.c_int32(deg), C.c_void_p(gy.data_ptr()),
                                            C.c_void_p(out.data_ptr()), C.c_int32(Cdim + 5), C.c_void_p(gx.data_ptr()),
                                            C.c_int32(1), _C._stream()))  # accumulate: twice the gradient
    assert rel_err(gx, 2 * xr.grad) <= 1e-4
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


@pytest.mark.parametrize('fused', [True, False])
def test_deform_mlp_gradient_reaches_the_joint_positions(fused):
    """joints are a trained parameter in stage sk (networks/sk_gs.py:507,607) and enter the network through the frequency
    encoding (sk_gs.py:1073): dL/d(joints) of both GPU paths equals torch autograd through reference_forward"""
    from sk_gs_amd.deform_net import DeformMLP
    torch.manual_seed(3)
    mlp = DeformMLP().cuda()
    joints = (torch.rand(20, 3, device='cuda') - 0.5).requires_grad_(True)
    t = torch.tensor([0.63], device='cuda')
    ref = torch.cat(mlp.reference_forward(joints, t), dim=-1)
    g = torch.randn_like(ref)
    ref.backward(g)
    g_joints, joints.grad = joints.grad.clone(), None
    gref = [p.grad.clone() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    mlp.force_layered = not fused
    out = torch.cat(mlp(joints, t), dim=-1)
    out.backward(g)
    assert float(g_joints.abs().max()) > 0
    # d(sin(2^9 x))/dx multiplies rounding of the argument by 512: looser than the weight gradients
    assert rel_err(joints.grad, g_joints) <= 2e-4
    for (n, p), gr in zip(mlp.named_parameters(), gref):
        assert rel_err(p.grad, gr) <= 5e-5, n


def _ref_with_input_grad(mlp, joints, t, g):
    """torch reference of forward + backward, including dL/dx0 (the encoded input as a leaf)"""
    from sk_gs_amd.deform_net import freq_encode_torch
    net = mlp.dynamic_net
    x0 = torch.cat([freq_encode_torch(joints, mlp.p_degree),
                    freq_encode_torch(t.view(-1, mlp.t_in), mlp.t_degree).expand(joints.shape[0], -1)], -1)
    x0 = x0.detach().requires_grad_(True)
    x, acts = x0, []
    for i in range(net.num_layers):
        x = F.relu(net.net[i](x))
        acts.append(x)
        if i in net.skips:
            x = torch.cat([x, x0], dim=-1)
    out = F.linear(x, net.last_weight, net.last_bias)
    for p in mlp.parameters():
        p.grad = None
    out.backward(g)
    params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
    return out.detach(), torch.stack([a.detach() for a in acts]), [p.grad.clone() for p in params], x0.grad.clone(), x0.detach()


def test_fused_deform_mlp_matches_torch():
    """the one-launch-per-direction network (csrc/mlp_fused.hip) against torch autograd of the restated module: outputs,
    saved activations, every weight / bias gradient and the input gradient, for row counts on both sides of the 16-row
    passes, repeated launches on one workspace (launch epochs) and a hipGraph replay"""
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    torch.manual_seed(0)
    mlp = DeformMLP().cuda()
    with torch.no_grad():  # the reference initialises the heads with std 1e-6 (sk_gs.py:542-545): use O(1) heads here
        mlp.dynamic_net.last_weight.normal_(0, 0.1)
    for B in (1, 3, 16, 20, 32, 33, 48):
        joints = (torch.rand(B, 3, device='cuda') - 0.5)
        t = torch.tensor([0.41], device='cuda')
        g = torch.randn(B, 11, device='cuda')
        ref_out, ref_acts, ref_grads, ref_gx0, ref_x0 = _ref_with_input_grad(mlp, joints, t, g)
        run = FusedDeformMLP(mlp, B)
        grads = [torch.full_like(r, 7.0) for r in ref_grads]
        g_x0 = torch.full_like(ref_gx0, 7.0)
        for rep in range(3):  # same workspace: the launch epoch distinguishes the exchanges
            out = run.forward(joints, t)
            run.backward(joints, t, g, grads, g_x0)
        st = run.status()
        assert (st['forward'], st['backward'], st['failed']) == (3, 3, 0)
        assert (run.x0 - ref_x0).abs().max() <= 3e-4  # sin of arguments up to 2^9 x: argument rounding
        assert rel_err(out, ref_out) <= 2e-5, B
        assert rel_err(run.acts, ref_acts) <= 2e-5, B
        for i, (a, r) in enumerate(zip(grads, ref_grads)):
            assert rel_err(a, r) <= 5e-5, (B, i)
        assert rel_err(g_x0, ref_gx0) <= 5e-5, B
        # without the input gradient nothing else changes; nor with the encoded input rebuilt inside the kernel
        for kw in (dict(), dict(reencode=True)):
            grads2 = [torch.zeros_like(r) for r in ref_grads]
            run.forward(joints, t)
            run.backward(joints, t, g, grads2, None, **kw)
            for i, (a, b) in enumerate(zip(grads, grads2)):
                if kw:  # sinf in the backward kernel vs the forward's: same code, same values
                    assert rel_err(a, b) <= 1e-6, (B, i)
                else:
                    assert torch.equal(a, b)
    # hipGraph replay of forward + backward (what the training step does)
    B = 20
    joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.7], device='cuda'), torch.randn(B, 11, device='cuda')
    ref_out, _, ref_grads, _, _ = _ref_with_input_grad(mlp, joints, t, g)
    run = FusedDeformMLP(mlp, B)
    grads = [torch.zeros_like(r) for r in ref_grads]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run.forward(joints, t)
        run.backward(joints, t, g, grads)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            run.forward(joints, t)
            run.backward(joints, t, g, grads)
        for _ in range(5):
            for gr in grads:
                gr.zero_()
            run.out.zero_()
            graph.replay()
    torch.cuda.synchronize()
    assert run.status()['failed'] == 0
    assert rel_err(run.out, ref_out) <= 2e-5
    for a, r in zip(grads, ref_grads):
        assert rel_err(a, r) <= 5e-5


@pytest.mark.parametrize('mode', [0, 1, 2, 3])
def test_fused_deform_mlp_where_the_network_runs(mode):
    """``skgs_deform_mlp_xcd_mode``: the network's 32 workgroups on blocks 0..31 (four per XCD, write-through exchange: 0), on blocks
    0, 8, .. 248 = one XCD with the exchange kept in that L2 when the launch's own census confirms the placement (1), that placement
    with write-through stores (2), and mode 1 with a census that says "two XCDs" (3: the fall-back every launch must be able to take).
    Same outputs and gradients in all of them (the arithmetic does not depend on the store flavour: bit-identical to mode 0), no launch
    gives up, with and without the optimizer's side job beside the network, eager and as a graph replay."""
    from sk_gs_amd import _C
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    lib = _C.load_library()
    lib.skgs_deform_mlp_xcd_mode.restype = C.c_int32
    before = lib.skgs_deform_mlp_xcd_mode(C.c_int32(-1))
    torch.manual_seed(1)
    mlp = DeformMLP().cuda()
    with torch.no_grad():
        mlp.dynamic_net.last_weight.normal_(0, 0.1)
    results = {}
    try:
        for m in (0, mode):
            assert lib.skgs_deform_mlp_xcd_mode(C.c_int32(m)) in (0, 1, 2, 3)
            for B in (20, 32, 48):
                torch.manual_seed(100 + B)
                joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.3], device='cuda'), torch.randn(B, 11, device='cuda')
                ref_out, _, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
                run = FusedDeformMLP(mlp, B)
                grads, g_x0 = [torch.zeros_like(r) for r in ref_grads], torch.zeros_like(ref_gx0)
                for rep in range(6):
                    out = run.forward(joints, t)
                    run.backward(joints, t, g, grads, g_x0)
                st = run.status()
                assert (st['forward'], st['backward'], st['failed']) == (6, 6, 0), (m, B, st)
                if m in (0, 2, 3):
                    assert st['one_xcd_forward'] == 0 and st['one_xcd_backward'] == 0, (m, st)
                else:  # every launch or none: the placement rule holds on this device or it does not (then mode 1 IS mode 2)
                    assert st['one_xcd_forward'] in (0, 6) and st['one_xcd_backward'] == st['one_xcd_forward'], st
                    print(f'[xcd] mode 1, B={B}: {st["one_xcd_forward"]} of 6 launches per direction on one XCD')
                assert rel_err(out, ref_out) <= 2e-5 and rel_err(g_x0, ref_gx0) <= 5e-5
                for a, r in zip(grads, ref_grads):
                    assert rel_err(a, r) <= 5e-5
                results[(m, B)] = [out.clone(), g_x0.clone()] + [x.clone() for x in grads]
        for B in (20, 32, 48):
            for a, b in zip(results[(0, B)], results[(mode, B)]):
                assert torch.equal(a, b), (mode, B)
    finally:
        lib.skgs_deform_mlp_xcd_mode(C.c_int32(before))


@pytest.mark.parametrize('mode', [0, 1])
def test_fused_deform_mlp_never_reads_an_earlier_launch(mode):
    """The in-launch exchange validates a word by its VALUE (anything but the sentinel): a word left over from an earlier launch of the
    same parity would pass as data.  The re-poisoning between launches and the kernel boundary's cache maintenance must therefore hold in
    every placement -- in mode 1 the slabs stay in ONE XCD's L2 and the network's XCD may change from launch to launch.  96 launches per
    direction on one workspace, fresh inputs AND fresh weights in every one (the working set is far too small to evict anything), each
    checked against torch: a stale slab would show as another launch's activations."""
    from sk_gs_amd import _C
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    lib = _C.load_library()
    lib.skgs_deform_mlp_xcd_mode.restype = C.c_int32
    before = lib.skgs_deform_mlp_xcd_mode(C.c_int32(mode))
    try:
        torch.manual_seed(5)
        mlp = DeformMLP().cuda()
        B = 20
        run = FusedDeformMLP(mlp, B)
        worst, ties = 0.0, 0
        shift = torch.zeros(8 * 1024, device='cuda')
        for it in range(96):
            with torch.no_grad():
                for p in mlp.parameters():
                    p.mul_(0.97).add_(torch.randn_like(p) * 0.02)  # (a bounded walk: the weights stay O(0.08))
                mlp.dynamic_net.last_weight.normal_(0, 0.1)
            joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.rand(1, device='cuda'), torch.randn(B, 11, device='cuda')
            ref_out, ref_acts, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
            grads, g_x0 = [torch.zeros_like(r) for r in ref_grads], torch.zeros_like(ref_gx0)
            # (launches of odd sizes in front of each direction: the dispatcher deals the next launch's blocks on from where the last one
            # stopped, so the network lands on different XCDs -- status()['xcds_*'] below)
            shift[: 1024 * (1 + it % 7)].add_(1.0)
            out = run.forward(joints, t)
            shift[: 1024 * (1 + (3 * it) % 5)].add_(1.0)
            run.backward(joints, t, g, grads, g_x0)
            errs = [rel_err(out, ref_out), rel_err(run.acts, ref_acts)]
            # (a pre-activation within rounding of zero may fall on the other side of the ReLU in torch's summation order: the two
            # backwards then differ by that unit's whole gradient -- a tie, not an error; such an iteration checks the forward only)
            tie = bool(((run.acts > 0) != (ref_acts > 0)).any())
            ties += tie
            if not tie:
                errs += [rel_err(g_x0, ref_gx0)] + [rel_err(a, r) for a, r in zip(grads, ref_grads)]
            worst = max(worst, max(float(e) for e in errs))
            assert worst <= 1e-4, (mode, it, errs)
        # the same through a captured graph (what the training step replays; its launches have been seen on another XCD than eager ones)
        joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.rand(1, device='cuda'), torch.randn(B, 11, device='cuda')
        ref_grads = _ref_with_input_grad(mlp, joints, t, g)[2]
        grads, g_x0 = [torch.zeros_like(r) for r in ref_grads], torch.zeros(B, mlp.dynamic_net.in_channels, device='cuda')
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            run.forward(joints, t)
            run.backward(joints, t, g, grads, g_x0)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                run.forward(joints, t)
                run.backward(joints, t, g, grads, g_x0)
            for it in range(64):
                with torch.no_grad():
                    for p in mlp.parameters():
                        p.mul_(0.97).add_(torch.randn_like(p) * 0.02)  # (a bounded walk: the weights stay O(0.08))
                    mlp.dynamic_net.last_weight.normal_(0, 0.1)
                    joints.copy_(torch.rand(B, 3, device='cuda') - 0.5), t.copy_(torch.rand(1, device='cuda')), g.copy_(torch.randn(B, 11, device='cuda'))
                ref_out, ref_acts, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
                if it % 2:
                    shift[: 1024 * (1 + it % 7)].add_(1.0)
                graph.replay()
                torch.cuda.synchronize()
                errs = [rel_err(run.out, ref_out), rel_err(run.acts, ref_acts)]
                tie = bool(((run.acts > 0) != (ref_acts > 0)).any())
                ties += tie
                if not tie:
                    errs += [rel_err(g_x0, ref_gx0)] + [rel_err(a, r) for a, r in zip(grads, ref_grads)]
                worst = max(worst, max(float(e) for e in errs))
                assert worst <= 1e-4, (mode, 'graph', it, errs)
            # ... and eager launches and replays TAKING TURNS: a launch returns to an XCD whose L2 last saw this parity's image two
            # launches ago, re-poisoned since then from the other XCD -- what it reads there must be the sentinel or this launch's slabs
            for it in range(48):
                with torch.no_grad():
                    for p in mlp.parameters():
                        p.mul_(0.97).add_(torch.randn_like(p) * 0.02)
                    mlp.dynamic_net.last_weight.normal_(0, 0.1)
                    joints.copy_(torch.rand(B, 3, device='cuda') - 0.5), t.copy_(torch.rand(1, device='cuda')), g.copy_(torch.randn(B, 11, device='cuda'))
                ref_out, ref_acts, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
                if it % 3 == 2:
                    graph.replay()
                else:
                    run.forward(joints, t)
                    run.backward(joints, t, g, grads, g_x0)
                torch.cuda.synchronize()
                errs = [rel_err(run.out, ref_out), rel_err(run.acts, ref_acts)]
                tie = bool(((run.acts > 0) != (ref_acts > 0)).any())
                ties += tie
                if not tie:
                    errs += [rel_err(g_x0, ref_gx0)] + [rel_err(a, r) for a, r in zip(grads, ref_grads)]
                worst = max(worst, max(float(e) for e in errs))
                assert worst <= 1e-4, (mode, 'turns', it, errs)
        st = run.status()
        assert (st['forward'], st['backward'], st['failed']) == (96 + 1 + 64 + 48, 96 + 1 + 64 + 48, 0)   # (the capture itself launches nothing)
        assert ties <= 24, ties
        print(f'[xcd] mode {mode}: 96 eager + 64 replayed + 48 alternating launches per direction with fresh inputs and weights ({ties} with a ReLU tie: forward only), worst relative error {worst:.2e}; '
              f'on one XCD: {st["one_xcd_forward"]} / {st["one_xcd_backward"]}; workgroup 0 ran on XCDs {st["xcds_forward"]} / {st["xcds_backward"]}')
    finally:
        lib.skgs_deform_mlp_xcd_mode(C.c_int32(before))


def test_fused_deform_mlp_other_shapes():
    """no skip, two skips, a skip right before the heads with another encoder; shapes outside the fused kernels' range
    are reported as unsupported (the per-layer path takes them)"""
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    torch.manual_seed(1)
    from sk_gs_amd.deform_net import fused_supported
    assert not fused_supported(DeformMLP(width=128), 20) and not fused_supported(DeformMLP(), 49)
    assert not fused_supported(DeformMLP(p_in_channels=4), 20)  # encoded width 97: not a multiple of 4
    for kw in (dict(depth=3, skips=()), dict(depth=4, skips=(1, 3)),
               dict(depth=2, skips=(1,), p_in_channels=5, p_degree=6, t_degree=3)):
        mlp = DeformMLP(**kw).cuda()
        B = 24
        joints = torch.rand(B, mlp.p_in, device='cuda') - 0.5
        t = torch.tensor([0.2], device='cuda')
        g = torch.randn(B, 11, device='cuda')
        ref_out, ref_acts, ref_grads, ref_gx0, _ = _ref_with_input_grad(mlp, joints, t, g)
        run = FusedDeformMLP(mlp, B)
        grads = [torch.zeros_like(r) for r in ref_grads]
        # (the input gradient is distributed over the workgroups by hidden column: it needs hidden >= encoded width)
        g_x0 = torch.zeros_like(ref_gx0) if mlp.dynamic_net.dim_hidden >= mlp.dynamic_net.in_channels else None
        out = run.forward(joints, t)
        run.backward(joints, t, g, grads, g_x0)
        assert run.status()['failed'] == 0
        assert rel_err(out, ref_out) <= 2e-5, kw
        for i, (a, r) in enumerate(zip(grads, ref_grads)):
            assert rel_err(a, r) <= 5e-5, (kw, i)
        assert g_x0 is None or rel_err(g_x0, ref_gx0) <= 5e-5, kw


def test_fused_deform_mlp_separate_heads():
    """the heads written to / read from separate tensors (sk_r | d_rot | d_scale, as the training step keeps them)"""
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    torch.manual_seed(3)
    mlp = DeformMLP().cuda()
    with torch.no_grad():
        mlp.dynamic_net.last_weight.normal_(0, 0.1)
    B = 20
    joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.9], device='cuda'), torch.randn(B, 11, device='cuda')
    run = FusedDeformMLP(mlp, B)
    ref_grads = [torch.zeros_like(p) for l in mlp.dynamic_net.net for p in (l.weight, l.bias)] + \
                [torch.zeros_like(mlp.dynamic_net.last_weight), torch.zeros_like(mlp.dynamic_net.last_bias)]
    out = run.forward(joints, t).clone()
    run.backward(joints, t, g, ref_grads)
    heads = [torch.zeros(B, 4, device='cuda'), torch.zeros(B, 4, device='cuda'), torch.zeros(B, 3, device='cuda')]
    run.forward(joints, t, head_out=heads)
    assert torch.equal(torch.cat(heads, dim=1), out)
    grads = [torch.zeros_like(r) for r in ref_grads]
    run.backward(joints, t, [h.contiguous() for h in g.split((4, 4, 3), dim=1)], grads)
    for a, r in zip(grads, ref_grads):
        assert torch.equal(a, r)


@pytest.mark.parametrize('M', [24, 40])
def test_skeleton_stage_in_one_launch_per_direction(M):
    """``skgs_skeleton_forward`` / ``skgs_skeleton_backward`` (network + kinematic chain in one launch) against the separate
    calls ``skgs_deform_mlp_*`` + ``skgs_bone_chain_*``: same bone transforms bit for bit (the same chain code on the same
    rotations); same gradients up to the order of the chain backward's LDS atomics"""
    from sk_gs_amd import _C
    from sk_gs_amd.deform_net import BoneChainDesc, FusedDeformMLP
    from sk_gs_amd.model import SkinnedGaussians
    torch.manual_seed(0)
    model = SkinnedGaussians(500, M, 4, sh_degree=0, num_frames=3, seed=5, deform_net=True, learn_joints=True).cuda()
    mlp, topo = model.sk_deform_net, model.topology()
    with torch.no_grad():
        mlp.dynamic_net.last_weight.normal_(0, 0.3)
        model.global_tr[1] = torch.tensor([0.1, -0.2, 0.05, 0.1, 0.2, -0.1, 0.9])
    joints, t = model.joints.detach().contiguous(), torch.tensor([0.37], device='cuda')
    gT = model.global_tr.detach()[1].contiguous()
    f32 = dict(dtype=torch.float32, device='cuda')
    net = mlp.dynamic_net
    params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]

    def heads():
        return [torch.empty((M, 4), **f32), torch.empty((M, 4), **f32), torch.empty((M, 3), **f32)]
    # ---- separate calls
    run = FusedDeformMLP(mlp, M)
    h1 = heads()
    run.forward(joints, t, head_out=h1)
    bone_T1, chain_A1 = _C.bone_chain_forward(h1[0], joints, gT, topo, save_chain=True)
    g_bone_T = torch.randn(M, 7, **f32)
    g_raw1, g_j1, g_g1 = _C.bone_chain_backward(h1[0], joints, gT, topo, chain_A1, g_bone_T, need_joints=True)
    gh = [g_raw1.contiguous(), torch.randn(M, 4, **f32), torch.randn(M, 3, **f32)]
    grads1, gx1 = [torch.zeros_like(p) for p in params], torch.zeros(M, net.in_channels, **f32)
    run.backward(joints, t, gh, grads1, gx1)
    # ---- one launch per direction
    b = BoneChainDesc()
    b.M, b.root, b.num_levels = M, topo['root'], topo['num_levels']
    b.parents, b.level_nodes, b.level_start = (topo['parents'].data_ptr(), topo['level_nodes'].data_ptr(),
                                                topo['level_start'].data_ptr())
    bone_T2, chain_A2 = torch.zeros(M, 7, **f32), torch.zeros(M, 7, **f32)
    g_j2, g_g2 = torch.zeros(M, 3, **f32), torch.zeros(7, **f32)
    h2 = heads()
    b.joints, b.global_T, b.bone_T, b.chain_A = joints.data_ptr(), gT.data_ptr(), bone_T2.data_ptr(), chain_A2.data_ptr()
    run2 = FusedDeformMLP(mlp, M)
    run2.forward(joints, t, head_out=h2, bones=b)
    for x, y in zip(h1, h2):
        assert torch.equal(x, y)
    assert torch.equal(bone_T1, bone_T2) and torch.equal(chain_A1, chain_A2)
    b.sk_r_raw, b.g_bone_T, b.g_joints, b.g_global_T = h2[0].data_ptr(), g_bone_T.data_ptr(), g_j2.data_ptr(), g_g2.data_ptr()
    gh2 = [torch.full((M, 4), 9.0, **f32), gh[1], gh[2]]  # head 0's gradient is an output here
    grads2, gx2 = [torch.zeros_like(p) for p in params], torch.zeros(M, net.in_channels, **f32)
    run2.backward(joints, t, gh2, grads2, gx2, bones=b)
    torch.cuda.synchronize()
    assert run2.status()['failed'] == 0
    assert rel_err(gh2[0], g_raw1) <= 1e-5 and rel_err(g_j2, g_j1) <= 1e-5 and rel_err(g_g2, g_g1) <= 1e-5
    assert rel_err(gx2, gx1) <= 1e-5
    for i, (x, y) in enumerate(zip(grads2, grads1)):
        assert rel_err(x, y) <= 1e-5, i


@pytest.mark.parametrize('M,learn_joints', [(20, True), (33, False)])
def test_operator_path_skeleton_stage_function(M, learn_joints):
    """``deform_net.skeleton_stage`` (what ``SkinnedGaussians.bone_transforms`` calls on the GPU: network + chain + cache row,
    one launch per direction under one autograd node) against (a) the composed operator path it replaces -- ``DeformMLP``'s
    Function, ``skeleton.bone_chain``, the torch expression of the cache row -- and (b) the plain-torch restatement
    (``reference_forward`` + ``skeleton.kinematic``): outputs, the cache row, and every gradient (network, joints through the
    chain AND through the network input, the frame's global transform; other frames' rows exactly zero)"""
    from sk_gs_amd import skeleton
    from sk_gs_amd.deform_net import skeleton_stage, skeleton_stage_supported
    from sk_gs_amd.model import SkinnedGaussians
    torch.manual_seed(1)
    frames, tid = 3, 1
    model = SkinnedGaussians(300, M, 4, sh_degree=0, num_frames=frames, seed=6, deform_net=True, learn_joints=learn_joints).cuda()
    mlp = model.sk_deform_net
    assert skeleton_stage_supported(mlp, M)
    with torch.no_grad():
        mlp.dynamic_net.last_weight.normal_(0, 0.3)
        model.global_tr[tid] = torch.tensor([0.1, -0.2, 0.05, 0.1, 0.2, -0.1, 0.9])
    g = torch.Generator().manual_seed(2)
    gT, gr, gs = (torch.randn(M, n, generator=g).cuda() for n in (7, 4, 3))
    params = dict(model.named_parameters())
    names = [n for n in params if n.startswith('sk_deform_net')] + ['global_tr'] + (['joints'] if learn_joints else [])

    def grads_of(fn):
        for p in model.parameters():
            p.grad = None
        model.sk_cache.zero_()
        bone_T, d_rot, d_scale = fn()
        ((bone_T * gT).sum() + (d_rot * gr).sum() + (d_scale * gs).sum()).backward()
        return (bone_T.detach(), d_rot.detach(), d_scale.detach(), model.sk_cache.clone(),
                {n: params[n].grad.clone() for n in names})

    def stage():
        return skeleton_stage(mlp, model.joints, model.frame_times[tid], model.global_tr, tid, model.topology(), model.sk_cache[tid])

    def composed():  # the operator path before: separate Functions + torch glue
        r, d_rot, d_scale = model.joint_outputs(tid)
        return skeleton.bone_chain(r, model.joints, model.global_tr[tid], model.topology()), d_rot, d_scale

    def plain_torch():
        r, d_rot, d_scale = mlp.reference_forward(model.joints, model.frame_times[tid])
        with torch.no_grad():
            model.sk_cache[tid] = torch.cat([F.normalize(r + model._rot_bias, dim=-1), d_rot, d_scale], dim=-1)
        sk_r = F.normalize(r + model._rot_bias, dim=-1)
        return skeleton.kinematic(model.joints, sk_r, model.global_tr[tid], model.joint_parents, model.joint_root,
                                  (model._ident7, model._root_mask)), d_rot, d_scale

    model.train()
    a, b, c = grads_of(stage), grads_of(composed), grads_of(plain_torch)
    for other, tol in ((b, 1e-5), (c, 2e-4)):
        for i in range(3):
            assert rel_err(a[i], other[i]) <= tol, i
        assert rel_err(a[3][tid], other[3][tid]) <= tol and float(a[3][[0, 2]].abs().max()) == 0.0
        for n in names:
            assert rel_err(a[4][n], other[4][n]) <= tol, n
    assert float(a[4]['global_tr'][[0, 2]].abs().max()) == 0.0
    # model.bone_transforms takes this path, and leaves the cache alone outside training
    model.sk_cache.zero_()
    with torch.no_grad():
        out = model.bone_transforms(tid)
    assert rel_err(out[0], a[0]) <= 1e-6 and float(model.sk_cache.abs().max()) == 0.0


def test_inference_forwards_reuse_one_fused_runner():
    """under no_grad no backward ever hands a runner (exchange workspace, activations) back: the forward returns it itself,
    so an inference loop (the FPS protocol) does not build and initialise a new one per call"""
    from sk_gs_amd import deform_net
    from sk_gs_amd.model import SkinnedGaussians
    model = SkinnedGaussians(200, 20, 4, sh_degree=0, num_frames=2, seed=2, deform_net=True, learn_joints=True).cuda()
    built = []
    orig = deform_net.FusedDeformMLP.__init__

    def counting(self, *a, **k):
        built.append(1)
        orig(self, *a, **k)
    deform_net.FusedDeformMLP.__init__ = counting
    try:
        with torch.no_grad():
            ref = model.bone_transforms(1)[0].clone()
            for _ in range(5):
                out = model.bone_transforms(1)[0]
                heads = model.sk_deform_net(model.joints, model.frame_times[1])
        assert len(built) == 1, built
        assert torch.equal(out, ref) and len(heads) == 3
        # with gradients on, a runner stays out until its backward has run
        a = model.bone_transforms(1)[0]
        b = model.bone_transforms(0)[0]
        (a.sum() + b.sum()).backward()
        assert len(built) == 2
        c = model.bone_transforms(1)[0]
        c.sum().backward()
        assert len(built) == 2
    finally:
        deform_net.FusedDeformMLP.__init__ = orig
