"""CPU tests of the two third-party stand-ins the reference's deform runs on (VERDICT r4, row b''): ``sk_gs_amd.lietorch`` (SE3 / SO3)
and ``sk_gs_amd.pytorch3d_ops`` (knn_points).

What pins them:
  * the reference's OWN code run on them: tests/golden/sk_stage.npz was written by ``SkeletonGaussianSplatting.forward`` of the
    unmodified networks/sk_gs.py (tests/golden/make_golden_sk_stage.py); ``test_reference_runs_unmodified_on_the_standins`` re-runs
    that in a child process (build container only) and compares with the committed file -- including the script's two checks that do
    not depend on the stand-ins' conventions (skeleton_warp_SE3 == the 4x4 twin skeleton_warp, values and gradients; the skinning ==
    the matrix branch of warp);
  * independent mathematics: fp64 finite differences of every group op in lietorch's LEFT-tangent convention (lie_cpu.cpp),
    scipy's rotation / matrix exponential for exp and log, torch.linalg.pinv for the closed-form projector inverse, plain torch
    autograd for the embedding gradient;
  * the C oracle (bone chain, search, skinning forward / backward) on the fixture's inputs.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from benchlib import ref_sequence as rs
from sk_gs_amd import lietorch as L
from sk_gs_amd import pytorch3d_ops as p3d
from sk_gs_amd.lietorch import SE3, SO3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
_REFERENCE = '/root/reference'


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rand_se3(n, g, dtype=torch.float64):
    v = torch.randn(n, 7, generator=g, dtype=dtype)
    v[:, 3:] = F.normalize(v[:, 3:], dim=-1)
    return v


# ------------------------------------------------------------------------------------------------ group mathematics
def test_exp_log_against_scipy():
    from scipy.linalg import expm
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(0)
    phi = torch.randn(64, 3, generator=g, dtype=torch.float64) * 1.2
    phi = phi * (2.8 / phi.norm(dim=-1, keepdim=True).clamp_min(2.8))        # log returns the rotation vector below pi
    phi[:4] *= 1e-9                                                     # the small-angle series (lie.h:149-152)
    q = SO3.exp(phi).vec()
    ref = torch.from_numpy(Rotation.from_rotvec(phi.numpy()).as_quat())  # xyzw
    ref = ref * torch.sign((ref * q).sum(-1, keepdim=True))
    assert (q - ref).abs().max() < 1e-12
    assert (SO3.exp(phi).log() - phi).abs().max() < 1e-9
    a = torch.randn(32, 6, generator=g, dtype=torch.float64) * 0.9
    a[:, 3:] = a[:, 3:] * (2.8 / a[:, 3:].norm(dim=-1, keepdim=True).clamp_min(2.8))
    a[:3, 3:] *= 1e-9
    X = SE3.exp(a)
    T = X.matrix()
    for i in range(a.shape[0]):
        tau, ph = a[i, :3].numpy(), a[i, 3:].numpy()
        H = np.zeros((4, 4))
        H[:3, :3] = [[0, -ph[2], ph[1]], [ph[2], 0, -ph[0]], [-ph[1], ph[0], 0]]
        H[:3, 3] = tau
        assert np.abs(expm(H) - T[i].numpy()).max() < 1e-12
    assert (X.log() - a).abs().max() < 1e-9
    assert (SE3.exp(a).inv() * SE3.exp(a)).log().abs().max() < 1e-12
    # the product is the matrix product, act is the matrix action
    Y = SE3.exp(torch.randn(32, 6, generator=g, dtype=torch.float64))
    assert ((X * Y).matrix() - X.matrix() @ Y.matrix()).abs().max() < 1e-12
    p = torch.randn(32, 3, generator=g, dtype=torch.float64)
    assert (X.act(p) - (X.matrix()[:, :3, :3] @ p[..., None])[..., 0] - X.matrix()[:, :3, 3]).abs().max() < 1e-12
    assert (X * p - X.act(p)).abs().max() == 0
    p4 = torch.cat([p, torch.full((32, 1), 2.0, dtype=torch.float64)], -1)
    assert (X.act(p4) - (X.matrix() @ p4[..., None])[..., 0]).abs().max() < 1e-12
    assert (X.translation()[:, :3] - X.matrix()[:, :3, 3]).abs().max() < 1e-12
    assert (SO3.exp(phi).matrix()[:, :3, :3] - torch.from_numpy(Rotation.from_rotvec(phi.numpy()).as_matrix())).abs().max() < 1e-12


def test_closed_form_projector_inverse_matches_linalg_pinv():
    """FromVec's backward is `grad @ pinv(orthogonal_projector)` upstream; here pinv is a closed form"""
    g = torch.Generator().manual_seed(1)
    v = _rand_se3(50, g)
    grad = torch.randn(50, 7, generator=g, dtype=torch.float64)
    grad[:, 6] = 0
    J = L.projector(SE3(v).group_id, v)
    assert (J[..., 6].abs().max() == 0) and J.shape == (50, 7, 7)
    want = (grad[:, None, :] @ torch.linalg.pinv(J))[:, 0]
    got = L._SE3Math.from_tangent(v, grad)
    assert _rel(got, want) < 1e-12
    q = v[:, 3:]
    Jq = L.projector(SO3(q).group_id, q)
    want = (grad[:, None, :4] @ torch.linalg.pinv(Jq))[:, 0]
    assert _rel(L._SO3Math.from_tangent(q, grad[:, :4]), want) < 1e-12
    # ToVec's backward: grad @ J
    e = torch.randn(50, 7, generator=g, dtype=torch.float64)
    assert _rel(L._SE3Math.to_tangent(v, e), (e[:, None, :] @ J)[:, 0]) < 1e-12


@pytest.mark.parametrize('group', ['SE3', 'SO3'])
def test_group_op_gradients_are_left_tangent_finite_differences(group):
    """lie_cpu.cpp's convention: the gradient of a group argument X is d/d(delta) f(Exp(delta) X) at 0, stored in the first K slots"""
    G = {'SE3': SE3, 'SO3': SO3}[group]
    K, N = G._math.K, G._math.N
    g = torch.Generator().manual_seed(2)
    n = 6
    X0 = G.exp(torch.randn(n, K, generator=g, dtype=torch.float64)).data
    Y0 = G.exp(torch.randn(n, K, generator=g, dtype=torch.float64)).data
    p = torch.randn(n, 3, generator=g, dtype=torch.float64)
    c3 = torch.randn(n, 3, generator=g, dtype=torch.float64)
    cK = torch.randn(n, K, generator=g, dtype=torch.float64)
    cN = torch.randn(n, N, generator=g, dtype=torch.float64)

    def scalar_of_group(Z):  # a smooth scalar of a group element that does not care about the sign of q
        return (Z.act(p) * c3).sum() + (Z.log() * cK).sum()

    cases = {
        'act': lambda X, Y: (X.act(p) * c3).sum(),
        'log': lambda X, Y: (X.log() * cK).sum(),
        'inv': lambda X, Y: scalar_of_group(X.inv()),
        'mul': lambda X, Y: scalar_of_group(X * Y),
        'vec': lambda X, Y: (X.vec() * cN).sum(),
        'matrix': lambda X, Y: (X.matrix()[..., :3, :] * torch.arange(12., dtype=torch.float64).view(3, 4)).sum(),
    }
    eps = 1e-6
    for name, f in cases.items():
        X, Y = X0.clone().requires_grad_(), Y0.clone().requires_grad_()
        f(G(X), G(Y)).backward()
        for which, leaf, other in (('X', X, Y0), ('Y', Y, X0)):
            if leaf.grad is None:
                continue
            assert leaf.grad[:, K:].abs().max() == 0, (name, 'slots beyond the tangent must be zero')
            fd = torch.zeros(n, K, dtype=torch.float64)
            for i in range(K):
                d = torch.zeros(n, K, dtype=torch.float64)
                d[:, i] = eps
                base = X0 if which == 'X' else Y0
                plus, minus = G(base).retr(d).data, G(base).retr(-d).data
                # per-row derivative: the scalar is a sum over rows, perturb all rows at once and difference row-wise via autograd-free eval
                for r in range(n):
                    bp, bm = base.clone(), base.clone()
                    bp[r], bm[r] = plus[r], minus[r]
                    args_p = (G(bp), G(other)) if which == 'X' else (G(other), G(bp))
                    args_m = (G(bm), G(other)) if which == 'X' else (G(other), G(bm))
                    fd[r, i] = (f(*args_p) - f(*args_m)) / (2 * eps)
            assert _rel(leaf.grad[:, :K], fd) < 2e-7, (group, name, which, _rel(leaf.grad[:, :K], fd))
    # exp: plain derivative w.r.t. the tangent vector
    a = torch.randn(n, K, generator=g, dtype=torch.float64).requires_grad_()
    torch.autograd.gradcheck(lambda t: scalar_of_group(G.exp(t)), (a,), eps=1e-6, atol=1e-6)
    # adj / adjT forward against the matrix
    A = G._math.Adj(X0)
    assert _rel(G(X0).adj(cK), (A @ cK[..., None])[..., 0]) < 1e-14 and _rel(G(X0).adjT(cK), (A.transpose(-1, -2) @ cK[..., None])[..., 0]) < 1e-14


def test_embedding_gradient_equals_plain_autograd():
    """for SE3.InitFromVec(v).act(p) lietorch's tangent gradient times pinv(projector) is the Euclidean gradient
    of the same function of v with the quaternion normalised inside (what the fused kernels return)"""
    g = torch.Generator().manual_seed(3)
    v = _rand_se3(40, g)
    p = torch.randn(40, 3, generator=g, dtype=torch.float64)
    c = torch.randn(40, 3, generator=g, dtype=torch.float64)
    a = v.clone().requires_grad_()
    (SE3.InitFromVec(a).act(p) * c).sum().backward()
    b = v.clone().requires_grad_()
    q = F.normalize(b[:, 3:], dim=-1)
    uv = 2 * torch.cross(q[:, :3], p, dim=-1)
    y = p + q[:, 3:] * uv + torch.cross(q[:, :3], uv, dim=-1) + b[:, :3]
    (y * c).sum().backward()
    assert _rel(a.grad, b.grad) < 1e-12


def test_interface_and_broadcasting():
    g = torch.Generator().manual_seed(4)
    X = SE3.InitFromVec(_rand_se3(6, g, torch.float32))
    assert X.shape == (6,) and X.tangent_shape == (6, 6) and X.data.shape == (6, 7) and len(X) == 6 and 'SE3' in repr(X)
    assert X[None].shape == (1, 6) and X[:, None].shape == (6, 1) and X[2].shape == () and X[2][None].shape == (1,)
    assert (X[None, :] * X[:, None]).shape == (6, 6)                       # same-rank broadcasting (lietorch/broadcasting.py)
    p = torch.randn(6, 4, 3, generator=g)
    assert X[:, None].act(p).shape == (6, 4, 3)
    assert (X[:, None].act(p)[3, 2] - X[3].act(p[3, 2])).abs().max() < 1e-6
    with pytest.raises(AssertionError):
        X.act(p)                                                            # ranks differ: upstream asserts too
    I = SE3.Identity(3, 2)
    assert I.shape == (3, 2) and (I.act(p[:3, :2]) - p[:3, :2]).abs().max() == 0
    assert SE3.IdentityLike(X).shape == (6,) and SO3.Random(5, generator=g).shape == (5,)
    assert (SE3(SO3(X)).data[:, :3] == 0).all() and (SO3(X).data == X.data[:, 3:]).all()
    assert X.view((2, 3)).shape == (2, 3) and X.detach().data.requires_grad is False
    assert L.cat([X, X], 0).shape == (12,) and L.stack([X, X], 0).shape == (2, 6)
    assert X.double().dtype == torch.float64 and X.to(torch.float64).dtype == torch.float64 and len(X.unbind(0)) == 6
    Y = SE3(X.data.clone())
    Y[1] = SE3.Identity(1)[0]
    assert (Y.data[1] == torch.tensor([0, 0, 0, 0, 0, 0, 1.])).all()
    assert (X.retr(torch.zeros(6, 6)).data - X.data).abs().max() < 1e-6
    assert (X.quaternion() - X.data[:, 3:]).abs().max() < 1e-6


# ------------------------------------------------------------------------------------------------ the deferred skinning expression
def test_skinning_expression_is_recognised_and_equals_the_generic_ops(monkeypatch):
    monkeypatch.setattr(L, '_FUSED_ON_CPU', True)
    g = torch.Generator().manual_seed(5)
    M, P, K = 20, 500, 5
    v = _rand_se3(M, g, torch.float32)
    idx = torch.randint(0, M, (P, K), generator=g)
    pts, w0 = torch.randn(P, 3, generator=g), torch.rand(P, K, generator=g)
    c = torch.randn(P, 3, generator=g)

    def run(fused):
        monkeypatch.setattr(L, '_FUSED', fused)
        a, w = v.clone().requires_grad_(), w0.clone().requires_grad_()
        T = SE3.InitFromVec(a)
        before = dict(L.fused_calls)
        d = (T[idx].act(pts[:, None]) * w[..., None]).sum(dim=1)
        (d * c).sum().backward()
        return d.detach(), a.grad, w.grad, {k: L.fused_calls[k] - before[k] for k in before}

    d1, ga1, gw1, n1 = run(True)
    d0, ga0, gw0, n0 = run(False)
    assert n1 == {'forward': 1, 'backward': 0, 'materialised': 0} and n0 == {'forward': 0, 'backward': 0, 'materialised': 0}
    assert _rel(d1, d0) < 1e-6 and _rel(ga1, ga0) < 1e-5 and _rel(gw1, gw0) < 1e-6
    # every other use of the product computes the real tensor: same values, same gradients
    monkeypatch.setattr(L, '_FUSED', True)
    T = SE3.InitFromVec(v)
    d = T[idx].act(pts[:, None])
    assert isinstance(d, torch.Tensor) and d.shape == (P, K, 3) and d.dtype == torch.float32 and d.dim() == 3 and d.numel() == P * K * 3
    before = L.fused_calls['materialised']
    assert _rel(d[7], SE3(v[idx[7]]).act(pts[7].expand(K, 3))) < 1e-6
    assert _rel((w0[..., None] * d).sum(1), d0) < 1e-6                          # reversed operands, positional dim
    assert _rel(torch.sum(d * w0[..., None], dim=-2), d0) < 1e-6
    assert _rel((d * w0[..., None]).sum(dim=1, keepdim=True)[:, 0], d0) < 1e-6     # keepdim: generic path
    assert _rel((d * 2.0).sum(dim=1), 2 * torch.stack([SE3(v[idx[:, k]]).act(pts) for k in range(K)], 1).sum(1)) < 1e-6
    assert L.fused_calls['materialised'] > before
    # K = 1 without weights: spT[p2sp].act(points) (warp method `largest`)
    p2sp = idx[:, 0].contiguous()
    assert _rel(T[p2sp].act(pts), SE3(v[p2sp]).act(pts)) < 1e-6
    # an index that is not a LongTensor [P] / [P,K] takes the plain gather
    assert type(T[1:3]) is SE3 and type(T[None]) is SE3


# ------------------------------------------------------------------------------------------------ the fixture
@pytest.fixture(scope='module')
def fixture():
    return np.load(os.path.join(GOLDEN, 'sk_stage.npz'))


@pytest.mark.parametrize('recognise', [False, True])
@pytest.mark.parametrize('name', sorted(rs.SCENARIOS))
def test_replay_of_the_reference_run(fixture, name, recognise, monkeypatch):
    """the restated call sequence (benchlib/ref_sequence.py) on the stand-ins reproduces what the reference's own forward + backward gave;
    `recognise`: with the deferred gather / skinning expression machinery of the HIP route switched on (evaluated by the generic ops here)"""
    monkeypatch.setattr(L, '_FUSED_ON_CPU', recognise)
    before = dict(L.fused_calls)
    res, out, got_grad, grad = rs.run_scenario(L, p3d.knn_points, fixture, name)
    assert L.fused_calls['forward'] - before['forward'] == int(recognise) and L.fused_calls['materialised'] == before['materialised']
    assert torch.equal(res['_knn_i'], out['_knn_i'])
    for k, want in out.items():
        if k != '_knn_i':
            assert _rel(res[k].detach().float() if res[k].is_floating_point() else res[k], want) < 2e-6, (name, k)
    assert set(grad) and all(got_grad[k] is not None for k in grad)
    for k, want in grad.items():
        assert _rel(got_grad[k], want) < 2e-5, (name, k, _rel(got_grad[k], want))


def test_replay_of_the_reference_arap_loss(fixture):
    """loss_sp_arap (sk_gs.py:1371-1381) as the reference itself ran it on the stand-ins: inv, product, log, act and their gradient"""
    spT = torch.from_numpy(fixture['arap/spT']).requires_grad_()
    loss, loss_ct = rs.loss_sp_arap(L, spT, torch.from_numpy(fixture['arap/sp_points']), int(fixture['arap/sk_knn_num']))
    (loss + 0.5 * loss_ct).backward()
    assert abs(float(loss) - float(fixture['arap/loss'])) < 2e-6 and abs(float(loss_ct) - float(fixture['arap/loss_ct'])) < 2e-6
    assert _rel(spT.grad, fixture['arap/g_spT']) < 2e-5


def test_fixture_against_the_oracle(fixture, oracle32):
    """sk_W: the C oracle's bone chain, search and skinning (forward and backward) on the reference run's inputs"""
    a, out, cot, grad = rs.load_scenario(fixture, 'sk_W')
    n = lambda t: t.detach().numpy()  # noqa: E731
    tid = int(a['time_id'])
    sk_r = F.normalize(a['net_sk_r'].detach() + torch.tensor([0., 0., 0., 1.]), dim=-1)                  # sk_gs.py:1076
    skT = oracle32.bone_chain_forward(n(a['parents_table']), int(a['root']), n(sk_r), n(a['joints']), n(a['global_tr'][tid]))
    sign = np.sign((skT[:, 3:] * n(out['_skT'])[:, 3:]).sum(-1, keepdims=True))     # q and -q are the same rotation
    assert np.abs(skT[:, :3] - n(out['_skT'])[:, :3]).max() < 2e-6 and np.abs(skT[:, 3:] * sign - n(out['_skT'])[:, 3:]).max() < 2e-6
    dist, idx = oracle32.knn_bones(n(a['_xyz']), n(a['joints']), 5)
    assert np.array_equal(idx, n(out['_knn_i']))
    fwd = oracle32.lbs_deform_forward(n(a['_xyz']), n(out['_knn_w']), idx, n(out['_skT']), n(a['net_d_rot']), n(a['net_d_scale']), n(a['_xyz']),
                                      n(a['_scaling']), n(a['_rotation']), n(a['_opacity']))
    for ok, k in (('means', 'points'), ('scales', 'scales'), ('rotations', 'rotations'), ('opacity', 'opacity'), ('d_xyz', '_d_xyz'),
                  ('d_rot', '_d_rot'), ('d_scale', '_d_scale')):
        assert _rel(fwd[ok], out[k]) < 2e-6, k
    # backward: the cotangents of the activated outputs and of the three blended deltas both reach the skinning
    z = lambda k: n(cot[k])  # noqa: E731
    bwd = oracle32.lbs_deform_backward(n(a['_xyz']), n(out['_knn_w']), idx, n(out['_skT']), n(a['net_d_rot']), n(a['net_d_scale']),
                                       n(a['_scaling']), n(a['_rotation']), n(a['_opacity']), z('points'), z('scales'), z('rotations'), z('opacity'))
    assert _rel(bwd['g_xyz'], grad['_xyz']) < 1e-5 and _rel(bwd['g_opacity_logit'], grad['_opacity']) < 1e-5
    assert _rel(bwd['g_log_scale'], grad['_scaling']) < 1e-5 and _rel(bwd['g_rot'], grad['_rotation']) < 1e-5


@pytest.mark.skipif(not os.path.isdir(_REFERENCE), reason='the reference is only mounted in the build container')
def test_reference_runs_unmodified_on_the_standins():
    """INTEGRATION.md section 4 run for real, in a child process: `install_as_lietorch()` + `install_as_pytorch3d()`, then the reference's
    own SkeletonGaussianSplatting (networks/sk_gs.py, unmodified) is constructed for the shipped configs' deform options and its
    forward(stage='sk' | 'sp') + backward reproduce tests/golden/sk_stage.npz; the script's checks against the reference's 4x4 twins run too."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden_sk_stage.py'), '--check'], capture_output=True, text=True, cwd='/tmp',
                       env=env, timeout=900)
    assert r.returncode == 0 and 'SK-STAGE-CHECK-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'skeleton_warp_SE3 vs skeleton_warp (4x4)' in r.stdout and 'skinning vs the 4x4 branch of warp' in r.stdout


# ------------------------------------------------------------------------------------------------ pytorch3d.ops
def test_knn_points_semantics():
    g = torch.Generator().manual_seed(6)
    p1, p2 = torch.randn(2, 40, 5, generator=g), torch.randn(2, 17, 5, generator=g)
    r = p3d.knn_points(p1, p2, K=4, return_nn=True)
    d = ((p1[:, :, None] - p2[:, None]) ** 2).sum(-1)
    want_d, want_i = torch.sort(d, dim=-1, stable=True)
    assert torch.equal(r.idx, want_i[..., :4]) and _rel(r.dists, want_d[..., :4]) < 1e-6 and r.idx.dtype == torch.int64
    assert torch.equal(r.knn, p3d.knn_gather(p2, r.idx)) and r.knn.shape == (2, 40, 4, 5)
    assert p3d.knn_points(p1, p2).idx.shape == (2, 40, 1) and p3d.knn_points(p1, p2, K=4).knn is None
    # ties: the lower index first
    q2 = torch.cat([p2, p2], dim=1)
    assert (p3d.knn_points(p1, q2, K=2).idx[..., 0] + 17 == p3d.knn_points(p1, q2, K=2).idx[..., 1]).all()
    # lengths, and K beyond the cloud: padded with zeros
    r = p3d.knn_points(p1, p2, lengths1=torch.tensor([40, 10]), lengths2=torch.tensor([17, 3]), K=5)
    assert (r.idx[1, :10, :3] < 3).all() and (r.idx[1, :, 3:] == 0).all() and (r.dists[1, :, 3:] == 0).all() and (r.dists[1, 10:] == 0).all()
    assert torch.equal(r.idx[0], want_i[0, :, :5])
    # L1
    r1 = p3d.knn_points(p1, p2, norm=1, K=3)
    assert _rel(r1.dists, torch.sort((p1[:, :, None] - p2[:, None]).abs().sum(-1), dim=-1).values[..., :3]) < 1e-6
    # gradient through the distances (pytorch3d's backward: 2 g (p1 - p2[idx]))
    a, b = p1[:1].clone().requires_grad_(), p2[:1].clone().requires_grad_()
    c = torch.randn(1, 40, 4, generator=g)
    (p3d.knn_points(a, b, K=4).dists * c).sum().backward()
    i = want_i[:1, :, :4]
    diff = p1[:1, :, None] - p2[0][i[0]][None]
    assert _rel(a.grad, (2 * c[..., None] * diff).sum(2)) < 1e-6
    gb = torch.zeros_like(p2[0]).index_add_(0, i.reshape(-1), (-2 * c[..., None] * diff).reshape(-1, 5))
    assert _rel(b.grad[0], gb) < 1e-6
    with pytest.raises(ValueError):
        p3d.knn_points(p1, p2[:1])


def test_knn_points_against_the_reference_fixture():
    """lbs_weights.npz: indices / distances of the reference's calc_LBS_weight runs (3-d and 3+8-d searches)"""
    z = np.load(os.path.join(GOLDEN, 'lbs_weights.npz'))
    n = 0
    for c, spec in enumerate(z['cases']):
        hyper = int(str(spec).split('|')[1])
        pts, sp = torch.from_numpy(z[f'c{c}.points']), torch.from_numpy(z[f'c{c}.sp_points'])
        if hyper:
            pts = torch.cat([pts, torch.from_numpy(z[f'c{c}.in.feature'])], -1)
            sp = torch.cat([sp, torch.from_numpy(z[f'c{c}.in.sp_feature'])], -1)
        idx = torch.from_numpy(z[f'c{c}.indices'])
        r = p3d.knn_points(pts[None], sp[None], None, None, K=idx.shape[1])
        assert torch.equal(r.idx[0], idx) and _rel(r.dists[0], z[f'c{c}.nn_dist']) < 2e-6
        n += 1
    assert n >= 8


def test_ball_query_semantics():
    g = torch.Generator().manual_seed(7)
    p1, p2 = torch.rand(1, 30, 3, generator=g), torch.rand(1, 50, 3, generator=g)
    r = p3d.ball_query(p1, p2, K=4, radius=0.3)
    d = ((p1[0, :, None] - p2[0][None]) ** 2).sum(-1)
    for n in range(30):
        want = torch.nonzero(d[n] < 0.09)[:, 0][:4]
        assert torch.equal(r.idx[0, n, :len(want)], want) and (r.idx[0, n, len(want):] == -1).all()
        assert _rel(r.dists[0, n, :len(want)], d[n, want]) < 1e-6 if len(want) else True
    assert r.knn.shape == (1, 30, 4, 3)


def test_install_hooks_plant_the_modules():
    import sk_gs_amd
    saved = {k: sys.modules.get(k) for k in ('lietorch', 'pytorch3d', 'pytorch3d.ops')}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        sk_gs_amd.install_as_lietorch()
        sk_gs_amd.install_as_pytorch3d()
        from lietorch import SE3 as A, SO3 as B  # noqa: F401
        from pytorch3d.ops import knn_points, ball_query  # noqa: F401
        import pytorch3d.ops
        assert A is SE3 and knn_points is p3d.knn_points and pytorch3d.ops.knn_points is p3d.knn_points
        sk_gs_amd.install_as_lietorch()                                      # idempotent
        sys.modules['lietorch'] = type(sys)('lietorch')
        with pytest.raises(RuntimeError):
            sk_gs_amd.install_as_lietorch()                                  # never shadows a real package
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
