"""GPU tests of the superpoint stage (stage `sp`, networks/sk_gs.py:830-856): the MFMA deform network (csrc/sp_mlp.hip) against
its torch restatement (itself pinned to the reference's DeformNetwork by tests/golden/sp_deformnet.npz), the 3+8-d search and
weightings against the oracle, and the fused step against the operator path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, to_np

pytestmark = pytest.mark.gpu


def _net(seed=0, heads=0.05):
    from sk_gs_amd.superpoint import SpDeformNet
    torch.manual_seed(seed)
    net = SpDeformNet()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # outputs and gradients of a readable size (reset_parameters leaves the heads at 1e-5 / 1e-8)
        for head in (net.gaussian_warp, net.gaussian_scaling, net.gaussian_rotation):
            head.weight.normal_(0, heads, generator=g)
            head.bias.normal_(0, 0.1, generator=g)
        for layer in net.linear:
            layer.bias.normal_(0, 0.05, generator=g)
        net.timenet[0].bias.normal_(0, 0.1, generator=g)
    return net.cuda()


@pytest.mark.parametrize('M', [512, 20, 100, 1000])
def test_sp_net_forward_matches_the_torch_restatement(M):
    net = _net(M)
    g = torch.Generator().manual_seed(M)
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    t = torch.tensor([0.3125], device='cuda')
    run = net.runner(M)
    run.forward(x, t)
    with torch.no_grad():
        ref = net.reference_forward(x, t)
    want = torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']], 1)
    assert rel_err(run.raw, want) <= 2e-5
    u = F.normalize(ref['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda'), dim=-1)
    assert rel_err(run.bone_T, torch.cat([ref['d_xyz'], u], 1)) <= 2e-5
    assert rel_err(run.d_rot, u) <= 2e-5 and rel_err(run.d_scale, ref['d_scaling']) <= 2e-5
    # the saved activations are the restatement's hidden state
    from sk_gs_amd.superpoint import _SpNetDesc  # noqa: F401
    Mp = (M + 15) // 16 * 16
    saved = run.saved.view(torch.float32)
    Y7 = saved[Mp * 96 + 7 * Mp * 256: Mp * 96 + 8 * Mp * 256].view(Mp, 256)[:M]
    assert rel_err(Y7, ref['hidden']) <= 2e-5


@pytest.mark.parametrize('M,stage', [(512, False), (512, True), (37, True), (1000, False)])
def test_sp_net_backward_matches_torch_autograd(M, stage):
    """every parameter gradient (8 layers, the time network, the three heads) of ONE row-block launch + ONE weight-gradient
    launch against torch autograd of the restatement; `stage`: cotangents w.r.t. (bone_T, d_rot, d_scale), the quaternion
    normalisation's backward inside the launch (sk_gs.py:847)"""
    net = _net(M + 1)
    g = torch.Generator().manual_seed(M + 7)
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    t = torch.tensor([0.7], device='cuda')
    run = net.runner(M)
    run.forward(x, t)
    ref = net.reference_forward(x, t)
    params = list(net.parameters())
    names = [n for n, _ in net.named_parameters()]
    if stage:
        g_T, g_r, g_s = (torch.randn(M, 7, generator=g).cuda(), torch.randn(M, 4, generator=g).cuda(),
                         torch.randn(M, 3, generator=g).cuda())
        u = F.normalize(ref['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda'), dim=-1)
        loss = (torch.cat([ref['d_xyz'], u], 1) * g_T).sum() + (u * g_r).sum() + (ref['d_scaling'] * g_s).sum()
    else:
        g_raw = torch.randn(M, 10, generator=g).cuda()
        loss = (torch.cat([ref['d_xyz'], ref['d_rotation'], ref['d_scaling']], 1) * g_raw).sum()
    want = torch.autograd.grad(loss, params)
    # the same gradients in fp64: a pre-activation within rounding distance of 0 takes the other side of the ReLU in ANY
    # two fp32 evaluations (torch's included) -- a parameter is held to max(5e-5, 3 x torch-fp32's own distance from fp64)
    import copy
    keep, net._runner = net._runner, None
    net64 = copy.deepcopy(net).double()
    net._runner = keep
    ref64 = net64.reference_forward(x.double(), t.double())
    if stage:
        u64 = F.normalize(ref64['d_rotation'] + torch.tensor([0, 0, 0, 1.], device='cuda', dtype=torch.float64), dim=-1)
        loss64 = ((torch.cat([ref64['d_xyz'], u64], 1) * g_T.double()).sum() + (u64 * g_r.double()).sum()
                  + (ref64['d_scaling'] * g_s.double()).sum())
    else:
        loss64 = (torch.cat([ref64['d_xyz'], ref64['d_rotation'], ref64['d_scaling']], 1) * g_raw.double()).sum()
    want64 = torch.autograd.grad(loss64, list(net64.parameters()))
    for p in params:
        p.grad = torch.full_like(p, float('nan'))  # written, not accumulated
    for rep in range(2):  # (twice: the ticket of the weight-gradient launch resets itself)
        if stage:
            run.backward(g_T, g_r, g_s)
        else:
            run.backward(None, None, None, g_raw=g_raw)
    torch.cuda.synchronize()
    for n, p, w, w64 in zip(names, params, want, want64):
        assert torch.isfinite(p.grad).all(), n
        tol = max(5e-5, 3.0 * rel_err(w.double(), w64))
        assert rel_err(p.grad.double(), w64) <= tol, (n, rel_err(p.grad.double(), w64), tol)


def test_sp_net_autograd_function_and_state_dict():
    """the operator path: SpDeformNet(x, t) as an autograd node; parameters carry the reference's state_dict names"""
    net = _net(3)
    assert {'timenet.0.weight', 'timenet.2.bias', 'linear.0.weight', 'linear.5.weight', 'gaussian_warp.weight',
            'gaussian_scaling.bias', 'gaussian_rotation.weight'} <= set(net.state_dict())
    assert tuple(net.linear[5].weight.shape) == (256, 349) and tuple(net.linear[0].weight.shape) == (256, 93)
    M = 512
    x = (torch.rand(M, 3) * 2 - 1).cuda()
    t = torch.tensor([0.25], device='cuda')
    out = net(x, t)
    ref = net.reference_forward(x, t)
    gs = [torch.randn_like(out[k]) for k in ('d_xyz', 'd_rotation', 'd_scaling')]
    got = torch.autograd.grad([out[k] for k in ('d_xyz', 'd_rotation', 'd_scaling')], list(net.parameters()), gs)
    want = torch.autograd.grad([ref[k] for k in ('d_xyz', 'd_rotation', 'd_scaling')], list(net.parameters()), gs)
    for (n, _), a, b in zip(net.named_parameters(), got, want):
        assert rel_err(a, b) <= 5e-5, n
    with pytest.raises(Exception):
        net(x.cpu(), t)
