"""GPU parity tests of the LBS deform, the bone KNN and the end-to-end model step against the CPU oracle.

Tolerances: KNN indices bit-exact; deform forward outputs <= 2e-6 max-norm relative (same operation order, only
expf differs by an ulp); deform backward <= 1e-5 (bone gradients are summed in a different order); end-to-end
parameter gradients of one training view <= 1e-4 (north star) with the threshold-flip allowance of helpers.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close_robust, rel_err, to_np

pytestmark = pytest.mark.gpu


def _inputs(P, M, K, seed):
    from sk_gs_amd import scene
    g = scene.make_gaussians(P, seed=seed)
    b = scene.make_bones(M, seed=seed)
    gen = torch.Generator().manual_seed(seed + 100)
    bone_T = torch.cat([0.3 * torch.randn(M, 3, generator=gen), torch.randn(M, 4, generator=gen)], -1)  # un-normalised q
    w = torch.softmax(torch.randn(P, K, generator=gen), -1)
    idx = torch.stack([torch.randperm(M, generator=gen)[:K] for _ in range(P)]).long()
    return g, b, bone_T, w, idx


@pytest.mark.parametrize('P,M,K', [(5000, 20, 5), (3000, 1, 1), (2500, 64, 8), (700, 65, 3), (2000, 600, 4), (1000, 1500, 3)])
def test_deform_forward_backward(oracle32, P, M, K):
    from sk_gs_amd import _C
    g, b, bone_T, w, idx = _inputs(P, M, K, seed=P)
    n = to_np
    ref = oracle32.lbs_deform_forward(n(g['xyz']), n(w), n(idx), n(bone_T), n(b['d_rot']), n(b['d_scale']), n(g['xyz']),
                                      n(g['log_scale']), n(g['rot']), n(g['opacity_logit']))
    c = lambda t: t.cuda()  # noqa: E731
    out = _C.lbs_deform_forward(c(g['xyz']), c(w), c(idx), c(bone_T), c(b['d_rot']), c(b['d_scale']), c(g['xyz']),
                                c(g['log_scale']), c(g['rot']), c(g['opacity_logit']), need_deltas=True)
    for name, t in zip(['means', 'scales', 'rotations', 'opacity', 'd_xyz', 'd_rot', 'd_scale'], out):
        assert rel_err(t, ref[name]) <= 2e-6, name
    gen = torch.Generator().manual_seed(1)
    gm, gs, gr, go = (torch.randn(P, 3, generator=gen), torch.randn(P, 3, generator=gen),
                      torch.randn(P, 4, generator=gen), torch.randn(P, 1, generator=gen))
    gref = oracle32.lbs_deform_backward(n(g['xyz']), n(w), n(idx), n(bone_T), n(b['d_rot']), n(b['d_scale']),
                                        n(g['log_scale']), n(g['rot']), n(g['opacity_logit']), n(gm), n(gs), n(gr), n(go))
    got = _C.lbs_deform_backward(c(g['xyz']), c(w), c(idx), c(bone_T), c(b['d_rot']), c(b['d_scale']),
                                 c(g['log_scale']), c(g['rot']), c(g['opacity_logit']), c(gm), c(gs), c(gr), c(go))
    names = ['g_weights', 'g_bone_T', 'g_bone_drot', 'g_bone_dscale', 'g_xyz', 'g_log_scale', 'g_rot',
             'g_opacity_logit']
    for name, t in zip(names, got):
        assert rel_err(t, gref[name]) <= 1e-5, (name, rel_err(t, gref[name]))


@pytest.mark.parametrize('P,M,K,dim', [(4000, 20, 5, 3), (1000, 512, 5, 11), (500, 3, 5, 3)])
def test_knn_bones(oracle32, P, M, K, dim):
    from sk_gs_amd import _C
    gen = torch.Generator().manual_seed(P + M)
    pts, jts = torch.randn(P, dim, generator=gen), torch.randn(M, dim, generator=gen)
    d_ref, i_ref = oracle32.knn_bones(to_np(pts), to_np(jts), K)
    d, i = _C.knn_bones(pts.cuda(), jts.cuda(), K)
    np.testing.assert_array_equal(to_np(i), i_ref)
    assert rel_err(d, d_ref) <= 1e-6


def test_autograd_op_matches_torch_reference():
    """lbs_deform (HIP, autograd.Function) vs the same math written with differentiable torch ops"""
    from sk_gs_amd import skeleton
    from sk_gs_amd.deform import lbs_deform
    P, M, K = 3000, 12, 4
    g, b, bone_T, w, idx = _inputs(P, M, K, seed=3)
    dev = 'cuda'
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in dict(
        w=w, bone_T=bone_T, drot=b['d_rot'], dscale=b['d_scale'], xyz=g['xyz'], ls=g['log_scale'], rot=g['rot'],
        op=g['opacity_logit']).items()}
    idx = idx.to(dev)

    def torch_ref(L):
        p = L['xyz'].detach()
        y = skeleton.se3_act(L['bone_T'][idx], p[:, None])
        d_xyz = (y * L['w'][..., None]).sum(1) - p
        d_rot = (L['drot'][idx] * L['w'][..., None]).sum(1)
        d_scale = (L['dscale'][idx] * L['w'][..., None]).sum(1)
        return (L['xyz'] + d_xyz, torch.exp(L['ls']) + d_scale, F.normalize(L['rot'] + d_rot, dim=-1),
                torch.sigmoid(L['op']))

    gen = torch.Generator().manual_seed(9)
    cot = [torch.randn(P, c, generator=gen).to(dev) for c in (3, 3, 4, 1)]
    outs_ref = torch_ref(leaves)
    grads_ref = torch.autograd.grad(outs_ref, list(leaves.values()), cot)
    outs = lbs_deform(leaves['xyz'].detach(), leaves['w'], idx, leaves['bone_T'], leaves['drot'], leaves['dscale'],
                      leaves['xyz'], leaves['ls'], leaves['rot'], leaves['op'])
    grads = torch.autograd.grad(outs, list(leaves.values()), cot)
    for a, r in zip(outs, outs_ref):
        assert rel_err(a, r) <= 1e-5
    for name, a, r in zip(leaves.keys(), grads, grads_ref):
        assert rel_err(a, r) <= 2e-5, (name, rel_err(a, r))


def test_model_step_matches_oracle(oracle32):
    """one full view: bone chain -> KNN weights -> deform -> render -> fixed cotangents -> grads of the six Gaussian
    parameter tensors, vs the oracle pipeline fed with the same bones"""
    from sk_gs_amd import scene
    from sk_gs_amd.model import SkinnedGaussians
    P, M, K, W, H = 6000, 10, 5, 160, 128
    model = SkinnedGaussians(P, M, K, num_frames=2, seed=1, scale_mult=2.5).cuda()
    cam = scene.make_camera(W, H, seed=4)
    rs = scene.raster_settings_from_camera(cam, colmap=True, device='cuda')
    out = model.render(rs, time_id=1)
    gen = torch.Generator().manual_seed(2)
    gcol, gop = torch.randn(3, H, W, generator=gen).cuda(), torch.randn(H, W, generator=gen).cuda()
    torch.autograd.backward([out['images'], out['opacity']], [gcol, gop])
    n = to_np
    with torch.no_grad():
        sk_T, d_rot, d_scale = model.bone_transforms(1)
        from sk_gs_amd.deform import calc_lbs_weight
        w, idx = calc_lbs_weight(model._xyz, model.joints, K, sp_W=model.sp_W)
        sh = torch.cat([model._features_dc, model._features_rest], 1)
    d = oracle32.lbs_deform_forward(n(model._xyz), n(w), n(idx), n(sk_T), n(d_rot), n(d_scale), n(model._xyz),
                                    n(model._scaling), n(model._rotation), n(model._opacity))
    fwd = oracle32.rasterize_forward(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix), n(rs.projmatrix),
                                     n(rs.campos), d['means'], d['opacity'], n(sh), d['scales'], d['rotations'])
    assert_close_robust(out['images'], fwd['color'], 1e-4, name='image')
    gr = oracle32.rasterize_backward(fwd, H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix),
                                     n(rs.projmatrix), n(rs.campos), d['means'], n(sh), d['scales'], d['rotations'],
                                     n(gcol), n(gop))
    gd = oracle32.lbs_deform_backward(n(model._xyz), n(w), n(idx), n(sk_T), n(d_rot), n(d_scale), n(model._scaling),
                                      n(model._rotation), n(model._opacity), gr['dL_dmeans3D'], gr['dL_dscales'],
                                      gr['dL_drotations'], gr['dL_dopacity'])
    assert_close_robust(model._xyz.grad, gd['g_xyz'], 1e-4, 1e-3, name='xyz')
    assert_close_robust(model._scaling.grad, gd['g_log_scale'], 1e-4, 1e-3, name='scaling')
    assert_close_robust(model._rotation.grad, gd['g_rot'], 1e-4, 1e-3, name='rotation')
    assert_close_robust(model._opacity.grad, gd['g_opacity_logit'], 1e-4, 1e-3, name='opacity')
    assert_close_robust(model._features_dc.grad, gr['dL_dsh'][:, :1], 1e-4, 1e-3, name='f_dc')
    assert_close_robust(model._features_rest.grad, gr['dL_dsh'][:, 1:], 1e-4, 1e-3, name='f_rest')
    # densification statistic: viewspace gradient populated
    vs = out['viewspace_points'].grad
    assert vs is not None
    assert_close_robust(vs, gr['dL_dmean2D'], 1e-4, 1e-3, name='viewspace grad')


@pytest.mark.parametrize('M,shape', [(20, 'random'), (64, 'chain'), (33, 'star'), (1, 'single'), (500, 'random')])
@pytest.mark.parametrize('with_global', [True, False])
def test_bone_chain_matches_torch(M, shape, with_global):
    """fused bone-chain kernel vs the torch restatement of kinematic + skeleton_warp_SE3 (values and gradients)"""
    from sk_gs_amd import skeleton
    gen = torch.Generator().manual_seed(M)
    if shape == 'chain':
        parents = torch.arange(-1, M - 1).clamp_min(0)
    elif shape == 'star':
        parents = torch.zeros(M, dtype=torch.long)
    else:
        parents = torch.zeros(M, dtype=torch.long)
        for i in range(1, M):
            parents[i] = int(torch.randint(0, i, (1,), generator=gen))
    joints = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
    raw = (0.3 * torch.randn(M, 4, generator=gen)).cuda().requires_grad_(True)
    gT = torch.cat([0.2 * torch.randn(3, generator=gen), torch.randn(4, generator=gen)]).cuda().requires_grad_(True)
    joints_g = joints.clone().requires_grad_(True)
    table, _ = skeleton.build_ancestor_table(parents, 0)
    table = table.cuda()
    consts = skeleton.root_constants(M, 0, 'cuda')
    bias = torch.tensor([0., 0., 0., 1.]).cuda()
    sk_r = F.normalize(raw + bias, dim=-1)
    ref = skeleton.kinematic(joints_g, sk_r, gT if with_global else None, table, 0, consts)
    cot = torch.randn(M, 7, generator=gen).cuda()
    # cotangent orthogonal to q, as the deform backward produces it
    qn = F.normalize(ref[:, 3:].detach(), dim=-1)
    cot[:, 3:] -= qn * (qn * cot[:, 3:]).sum(-1, keepdim=True)
    inputs = [raw, joints_g] + ([gT] if with_global else [])
    gref = torch.autograd.grad(ref, inputs, cot)
    topo = skeleton.build_topology(parents, 0, device='cuda')
    out = skeleton.bone_chain(raw, joints_g, gT if with_global else None, topo)
    got = torch.autograd.grad(out, inputs, cot)
    assert rel_err(out, ref) <= 5e-6
    for a, r in zip(got, gref):
        assert rel_err(a, r) <= 5e-5, rel_err(a, r)


def test_densify_stats_and_lbs_weights_kernels(oracle32):
    """(f)-4 densification statistics and the sp_W weighting against their numpy restatements"""
    from sk_gs_amd import _C
    g = torch.Generator().manual_seed(9)
    P, M, K = 5000, 12, 4
    radii = (torch.randint(-2, 30, (P,), generator=g)).clamp(min=0).int()
    grad = torch.randn(P, 3, generator=g)
    acc, den, mr = torch.rand(P, 1, generator=g), torch.randint(0, 5, (P, 1), generator=g).float(), torch.rand(P, generator=g) * 20
    ref = oracle32.densify_stats(radii.numpy(), grad.numpy(), acc.numpy(), den.numpy(), mr.numpy())
    a, d, m = acc.cuda(), den.cuda(), mr.cuda()
    _C.densify_stats(radii.cuda(), grad.cuda(), a, d, m)
    assert rel_err(a, ref[0]) <= 1e-6 and np.array_equal(to_np(d), ref[1]) and np.array_equal(to_np(m), ref[2])
    # KNN + softmax of the gathered logits in one launch vs knn_bones + numpy
    pts, joints, sp_W = torch.randn(P, 3, generator=g), torch.randn(M, 3, generator=g), torch.randn(P, M, generator=g)
    lib = _C.load_library()
    idx = torch.empty((P, K), dtype=torch.int64, device='cuda')
    w = torch.empty((P, K), device='cuda')
    import ctypes as C
    pts_d, joints_d, sp_W_d = pts.cuda(), joints.cuda(), sp_W.cuda()
    _C._check(lib.skgs_knn_lbs_weights(C.c_int32(P), C.c_int32(M), C.c_int32(K), C.c_void_p(pts_d.data_ptr()),
                                       C.c_void_p(joints_d.data_ptr()), C.c_void_p(sp_W_d.data_ptr()),
                                       C.c_void_p(idx.data_ptr()), C.c_void_p(w.data_ptr()), _C._stream()))
    torch.cuda.synchronize()
    _, idx_ref = oracle32.knn_bones(pts.numpy(), joints.numpy(), K)
    assert np.array_equal(to_np(idx), idx_ref)
    assert rel_err(w, oracle32.lbs_weights(sp_W.numpy(), idx_ref)) <= 2e-6


@pytest.mark.parametrize('P,M,K', [(5000, 20, 5), (3001, 7, 4), (900, 33, 8), (400, 3, 1)])
def test_knn_weights_deform_in_one_launch_is_bit_identical_to_the_separate_calls(oracle32, P, M, K):
    """skgs_knn_lbs_deform_forward == skgs_knn_lbs_weights followed by skgs_lbs_deform_forward (and the oracle)"""
    import ctypes as C
    from sk_gs_amd import _C
    g, b, bone_T, _, _ = _inputs(P, M, K, seed=P + 1)
    gen = torch.Generator().manual_seed(P)
    joints, sp_W = torch.randn(M, 3, generator=gen), torch.randn(P, M, generator=gen)
    d = {k: v.cuda().contiguous() for k, v in dict(xyz=g['xyz'], ls=g['log_scale'], rot=g['rot'], op=g['opacity_logit'],
                                                   joints=joints, sp_W=sp_W, bone_T=bone_T, drot=b['d_rot'],
                                                   dscale=b['d_scale']).items()}
    lib, st, p = _C.load_library(), _C._stream(), (lambda t: C.c_void_p(t.data_ptr()))
    idx1 = torch.empty((P, K), dtype=torch.int64, device='cuda')
    w1 = torch.empty((P, K), device='cuda')
    _C._check(lib.skgs_knn_lbs_weights(C.c_int32(P), C.c_int32(M), C.c_int32(K), p(d['xyz']), p(d['joints']), p(d['sp_W']),
                                       p(idx1), p(w1), st))
    sep = _C.lbs_deform_forward(d['xyz'], w1, idx1, d['bone_T'], d['drot'], d['dscale'], d['xyz'], d['ls'], d['rot'], d['op'])
    idx2, w2 = torch.empty_like(idx1), torch.empty_like(w1)
    outs = [torch.empty((P, c), device='cuda') for c in (3, 3, 4, 1)]
    _C._check(lib.skgs_knn_lbs_deform_forward(
        C.c_int32(P), C.c_int32(M), C.c_int32(K), p(d['xyz']), p(d['joints']), p(d['sp_W']), p(d['bone_T']), p(d['drot']),
        p(d['dscale']), p(d['xyz']), p(d['ls']), p(d['rot']), p(d['op']), p(idx2), p(w2), *[p(o) for o in outs], None, st))
    torch.cuda.synchronize()
    assert torch.equal(idx1, idx2) and torch.equal(w1, w2)
    for name, a, r in zip(['means', 'scales', 'rotations', 'opacity'], outs, sep):
        assert torch.equal(a, r.view_as(a)), name
    n = to_np
    _, idx_ref = oracle32.knn_bones(n(g['xyz']), n(joints), K)
    w_ref = oracle32.lbs_weights(n(sp_W), idx_ref)
    ref = oracle32.lbs_deform_forward(n(g['xyz']), w_ref, idx_ref, n(bone_T), n(b['d_rot']), n(b['d_scale']), n(g['xyz']),
                                      n(g['log_scale']), n(g['rot']), n(g['opacity_logit']))
    assert np.array_equal(n(idx2), idx_ref)
    for name, a in zip(['means', 'scales', 'rotations', 'opacity'], outs):
        assert rel_err(a, ref[name]) <= 3e-6, name


@pytest.mark.parametrize('P,M,K', [(5000, 20, 5), (3001, 7, 4), (700, 60, 8), (300, 3, 1)])
def test_deform_backward_with_logit_gradient_folded_in(P, M, K):
    """skgs_lbs_deform_backward_logits == skgs_lbs_deform_backward + skgs_lbs_weights_backward(_compact), bit for bit"""
    import ctypes as C
    from sk_gs_amd import _C
    g, b, bone_T, w, idx = _inputs(P, M, K, seed=P + 2)
    gen = torch.Generator().manual_seed(P)
    up = [torch.randn(P, c, generator=gen).cuda() for c in (3, 3, 4, 1)]
    t = {k: v.cuda().contiguous() for k, v in dict(xyz=g['xyz'], ls=g['log_scale'], rot=g['rot'], op=g['opacity_logit'],
                                                   w=w, idx=idx, bone_T=bone_T, drot=b['d_rot'], dscale=b['d_scale']).items()}
    lib, st, p = _C.load_library(), _C._stream(), (lambda x: None if x is None else C.c_void_p(x.data_ptr()))
    d = _C._DeformInputs()
    d.P, d.K, d.M = P, K, M
    d.points = d.xyz = t['xyz'].data_ptr()
    d.weights, d.indices, d.bone_T = t['w'].data_ptr(), t['idx'].data_ptr(), t['bone_T'].data_ptr()
    d.bone_drot, d.bone_dscale = t['drot'].data_ptr(), t['dscale'].data_ptr()
    d.log_scale, d.rot, d.opacity_logit = t['ls'].data_ptr(), t['rot'].data_ptr(), t['op'].data_ptr()
    ws = torch.empty((lib.skgs_lbs_deform_backward_workspace_bytes(C.c_int32(P), C.c_int32(M)),), dtype=torch.uint8, device='cuda')

    def outs():
        return dict(gw=torch.empty((P, K), device='cuda'), gT=torch.empty((M, 7), device='cuda'),
                    gdr=torch.empty((M, 4), device='cuda'), gds=torch.empty((M, 3), device='cuda'),
                    gx=torch.empty((P, 3), device='cuda'), gls=torch.empty((P, 3), device='cuda'),
                    grot=torch.empty((P, 4), device='cuda'), gop=torch.empty((P, 1), device='cuda'))

    a = outs()
    _C._check(lib.skgs_lbs_deform_backward(C.byref(d), *[p(x) for x in up], p(a['gw']), p(a['gT']), p(a['gdr']), p(a['gds']),
                                           p(a['gx']), p(a['gls']), p(a['grot']), p(a['gop']), p(ws), C.c_size_t(ws.numel()), st))
    dense_ref = torch.empty((P, M), device='cuda')
    compact_ref = torch.empty((P, K), device='cuda')
    _C._check(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(M), C.c_int32(K), p(t['w']), p(t['idx']), p(a['gw']),
                                            p(dense_ref), st))
    _C._check(lib.skgs_lbs_weights_backward_compact(C.c_int32(P), C.c_int32(K), p(t['w']), p(a['gw']), p(compact_ref), st))
    for dense_out, compact_out, with_gw in ((True, False, False), (False, True, True), (True, True, True)):
        o = outs()
        dense = torch.full((P, M), float('nan'), device='cuda') if dense_out else None
        compact = torch.full((P, K), float('nan'), device='cuda') if compact_out else None
        _C._check(lib.skgs_lbs_deform_backward_logits(
            C.byref(d), *[p(x) for x in up], p(o['gw']) if with_gw else None, p(o['gT']), p(o['gdr']), p(o['gds']), p(o['gx']),
            p(o['gls']), p(o['grot']), p(o['gop']), p(dense), p(compact), p(ws), C.c_size_t(ws.numel()), st))
        torch.cuda.synchronize()
        for k in o:
            if k != 'gw' or with_gw:
                assert torch.equal(o[k], a[k]), k
        if dense_out:
            assert torch.equal(dense, dense_ref)
        if compact_out:
            assert torch.equal(compact, compact_ref)
    # limits are reported, not silently exceeded
    d.K = 9
    assert lib.skgs_lbs_deform_backward_logits(C.byref(d), *[p(x) for x in up], None, p(a['gT']), p(a['gdr']), p(a['gds']),
                                               p(a['gx']), p(a['gls']), p(a['grot']), p(a['gop']), p(dense_ref), None, p(ws),
                                               C.c_size_t(ws.numel()), st) != 0


def test_calc_lbs_weight_W_method_one_launch_matches_torch():
    """calc_lbs_weight(sp_W=...) (one launch per direction) vs knn + torch gather / softmax and its autograd"""
    from sk_gs_amd import _C
    from sk_gs_amd.deform import calc_lbs_weight
    g = torch.Generator().manual_seed(11)
    P, M, K = 4000, 20, 5
    pts, joints = torch.randn(P, 3, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    sp_W = torch.randn(P, M, generator=g).cuda().requires_grad_(True)
    w, idx = calc_lbs_weight(pts, joints, K, sp_W=sp_W)
    _, idx_ref = _C.knn_bones(pts, joints, K)
    ref_W = sp_W.detach().clone().requires_grad_(True)
    w_ref = torch.gather(ref_W, 1, idx_ref).softmax(-1)
    assert torch.equal(idx, idx_ref) and rel_err(w, w_ref) <= 2e-6
    cot = torch.randn(P, K, generator=g).cuda()
    (g1,) = torch.autograd.grad(w, sp_W, cot)
    (g2,) = torch.autograd.grad(w_ref, ref_W, cot)
    assert rel_err(g1, g2) <= 2e-6


@pytest.mark.parametrize('P,M,K,dim', [(4000, 20, 5, 3), (1500, 512, 5, 11)])
def test_kernel_and_dist_lbs_weightings_match_oracle(oracle32, P, M, K, dim):
    """the two distance-based branches of calc_LBS_weight (networks/sk_gs.py:759-766 `weighted_kernel` / `kernel`,
    :769-770 `dist`), on xyz and on the 11-dimensional [xyz | hyper feature] space of the sp stage (:753-755): HIP KNN +
    torch glue against the oracle's restatement, forward and the gradients autograd hands to the joints, radii and
    weights"""
    from sk_gs_amd.deform import calc_lbs_weight
    gen = torch.Generator().manual_seed(P + M)
    pts3, jts3 = torch.randn(P, 3, generator=gen), torch.randn(M, 3, generator=gen)
    feat, jfeat = (torch.randn(P, dim - 3, generator=gen), torch.randn(M, dim - 3, generator=gen)) if dim > 3 else (None, None)
    radius = torch.exp(0.3 * torch.randn(M, generator=gen))
    kweight = torch.sigmoid(torch.randn(M, generator=gen))
    g_w = torch.randn(P, K, generator=gen)
    c = lambda t: None if t is None else t.cuda()  # noqa: E731
    full_p = pts3 if dim == 3 else torch.cat([pts3, feat], -1)
    full_j = jts3 if dim == 3 else torch.cat([jts3, jfeat], -1)
    d_ref, i_ref = oracle32.knn_bones(to_np(full_p), to_np(full_j), K)
    for method in ('weighted_kernel', 'kernel', 'dist'):
        jt = c(jts3).requires_grad_(True)
        ft, jft = (c(feat).requires_grad_(True), c(jfeat).requires_grad_(True)) if dim > 3 else (None, None)
        rad, kw = c(radius).requires_grad_(True), c(kweight).requires_grad_(True)
        kwargs = dict(kernel_radius=rad, kernel_weight=kw if method == 'weighted_kernel' else None) if method != 'dist' \
            else dict(temperature=0.7)
        w, idx = calc_lbs_weight(c(pts3), jt, K, feature=ft, sp_feature=jft, **kwargs)
        np.testing.assert_array_equal(to_np(idx), i_ref)
        (w * c(g_w)).sum().backward()
        if method == 'dist':
            w_ref, g_dist = oracle32.lbs_weights_dist(d_ref, 0.7, to_np(g_w))
        else:
            w_ref, gr = oracle32.lbs_weights_kernel(d_ref, i_ref, to_np(radius), to_np(kweight) if method == 'weighted_kernel'
                                                    else None, to_np(g_w))
            g_dist = gr['g_dist']
            assert rel_err(rad.grad, gr['g_radius']) <= 2e-5, method
            if method == 'weighted_kernel':
                assert rel_err(kw.grad, gr['g_weight']) <= 2e-5, method
        assert rel_err(w, w_ref) <= 2e-6, method
        g_p, g_j = oracle32.knn_dist_backward(to_np(full_p), to_np(full_j), i_ref, g_dist)
        if dim == 3:
            assert rel_err(jt.grad, g_j) <= 2e-5, method
        else:  # the xyz parts are detached when features are concatenated (sk_gs.py:754-755): only the features learn
            assert jt.grad is None
            assert rel_err(jft.grad, g_j[:, 3:]) <= 2e-5, method
            assert rel_err(ft.grad, g_p[:, 3:]) <= 2e-5, method


@pytest.mark.parametrize('cfg,P,M', [(1, 100_000, 20), (2, 200_000, 32), (3, 300_000, 20), (4, 500_000, 24)])
def test_deform_and_knn_at_full_size(oracle32, cfg, P, M):
    """the deform / KNN / LBS-weight kernels at BASELINE.json's full Gaussian and bone counts against the oracle (seconds
    of CPU): KNN indices and softmax weights, the fused KNN + weights + skinning launch, and the skinning backward"""
    from sk_gs_amd import _C, scene
    K = 5
    g = scene.make_gaussians(P, seed=cfg)
    b = scene.make_bones(M, seed=cfg)
    gen = torch.Generator().manual_seed(cfg + 100)
    bone_T = torch.cat([0.3 * torch.randn(M, 3, generator=gen), torch.randn(M, 4, generator=gen)], -1)
    sp_W = torch.randn(P, M, generator=gen)
    n, c = to_np, (lambda t: t.cuda())
    d_ref, i_ref = oracle32.knn_bones(n(g['xyz']), n(b['joints']), K)
    w_ref = oracle32.lbs_weights(n(sp_W), i_ref)
    dist, idx = _C.knn_bones(c(g['xyz']), c(b['joints']), K)
    np.testing.assert_array_equal(n(idx), i_ref)
    assert rel_err(dist, d_ref) <= 1e-6
    ref = oracle32.lbs_deform_forward(n(g['xyz']), w_ref, i_ref, n(bone_T), n(b['d_rot']), n(b['d_scale']), n(g['xyz']),
                                      n(g['log_scale']), n(g['rot']), n(g['opacity_logit']))
    # the one-launch path of the training step: search + softmax + skinning
    lib = _C.load_library()
    import ctypes as C
    f32 = dict(dtype=torch.float32, device='cuda')
    o_idx, o_w = torch.empty((P, K), dtype=torch.int64, device='cuda'), torch.empty((P, K), **f32)
    means, scales, rots, opac = (torch.empty((P, 3), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32),
                                 torch.empty((P, 1), **f32))
    dev = {k: c(v) for k, v in dict(xyz=g['xyz'], ls=g['log_scale'], rot=g['rot'], op=g['opacity_logit'], j=b['joints'],
                                    spw=sp_W, bT=bone_T, dr=b['d_rot'], ds=b['d_scale']).items()}
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _C._check(lib.skgs_knn_lbs_deform_forward(
        C.c_int32(P), C.c_int32(M), C.c_int32(K), p(dev['xyz']), p(dev['j']), p(dev['spw']), p(dev['bT']), p(dev['dr']),
        p(dev['ds']), p(dev['xyz']), p(dev['ls']), p(dev['rot']), p(dev['op']), p(o_idx), p(o_w), p(means), p(scales),
        p(rots), p(opac), None, _C._stream()))
    np.testing.assert_array_equal(n(o_idx), i_ref)
    assert rel_err(o_w, w_ref) <= 2e-6
    for name, t in zip(['means', 'scales', 'rotations', 'opacity'], (means, scales, rots, opac)):
        assert rel_err(t, ref[name]) <= 2e-6, name
    gm, gs, gr, go = (torch.randn(P, 3, generator=gen), torch.randn(P, 3, generator=gen),
                      torch.randn(P, 4, generator=gen), torch.randn(P, 1, generator=gen))
    gref = oracle32.lbs_deform_backward(n(g['xyz']), w_ref, i_ref, n(bone_T), n(b['d_rot']), n(b['d_scale']),
                                        n(g['log_scale']), n(g['rot']), n(g['opacity_logit']), n(gm), n(gs), n(gr), n(go))
    got = _C.lbs_deform_backward(dev['xyz'], o_w, o_idx, dev['bT'], dev['dr'], dev['ds'], dev['ls'], dev['rot'], dev['op'],
                                 c(gm), c(gs), c(gr), c(go))
    names = ['g_weights', 'g_bone_T', 'g_bone_drot', 'g_bone_dscale', 'g_xyz', 'g_log_scale', 'g_rot', 'g_opacity_logit']
    for name, t in zip(names, got):
        # per-bone gradients sum 10^5..10^6 terms in fp32 in two different orders: 1e-4; per-Gaussian ones 1e-5
        assert rel_err(t, gref[name]) <= (1e-4 if 'bone' in name else 1e-5), (name, rel_err(t, gref[name]))


def test_inference_forward_takes_the_one_launch_skinning_and_matches_the_training_forward():
    """under no_grad ``SkinnedGaussians.forward`` runs search + weights + skinning as the fused step's ONE launch
    (``_C.knn_lbs_deform_forward``); with gradients on, the two autograd Functions: same outputs bit for bit"""
    from sk_gs_amd.model import SkinnedGaussians
    model = SkinnedGaussians(5000, 20, 5, sh_degree=1, num_frames=2, seed=4, deform_net=True).cuda()
    with torch.no_grad():
        a = model(1)
    b = model(1)
    assert b['points'].requires_grad and not a['points'].requires_grad
    for k in ('points', 'scales', 'rotations', 'opacity'):
        assert torch.equal(a[k], b[k].detach()), k
