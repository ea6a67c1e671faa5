"""The fused step UNDER the reference's own API (sk_gs_amd/reference_fused.py): ``SkeletonGaussianSplatting.render`` +
``ImageLoss.forward`` + ``SSIM_Loss.forward`` as ``accelerate_reference()`` patches them, driven on a stand-in model with the reference's
attribute names (benchlib/reference_loop.py; the real class runs the same condition / re-homing code in tests/test_host_cpu.py) against
the reference's own call sequence on the stand-ins (benchlib/ref_sequence.py, pinned by tests/golden/sk_stage.npz)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = {9: dict(name='small-4k-160', P=4000, M=12, K=4, W=160, H=120)}


def _setup(mode, extra=()):
    from benchlib import options, reference_loop
    from sk_gs_amd import reference_accel as ra, reference_fused as rf
    ra.restore_reference()
    for k in list(ra._originals):
        ra._originals.pop(k)
    for k in rf.calls:
        rf.calls[k] = 0
    args = options.build_parser().parse_args(['--reference-loop', mode, '--config', '9', '--views', '3', '--scale-mult', '2.0', *extra])
    return reference_loop.setup(args, SMALL)


def _teardown(s):
    import torch.optim
    if 'adam' in s.ra._originals:
        torch.optim.Adam.step = s.ra._originals.pop('adam')
    for k in list(s.ra._originals):
        s.ra._originals.pop(k)


def test_one_iteration_gives_the_gradients_of_the_reference_sequence():
    """forward values and EVERY parameter's gradient of one iteration through the fused route == the reference's own sequence (stand-ins +
    rasterizer adapter + torch L1 + fused SSIM) on the same parameters; the returned dict holds what ``loss`` / ``adaptive_control`` read"""
    s = _setup('fused')
    try:
        rf, v = s.rf, 1
        names = dict(s.p)
        names.update({f'net.{n}': q for n, q in s.net.named_parameters()})
        # reference sequence (the `accelerated` path of the same loop) -> autograd gradients
        for q in names.values():
            q.grad = None
        img = s.render(v, s.deform(v))
        loss_ref = s.loss_of(img, s.targets[v])
        loss_ref.backward()
        want = {n: (None if q.grad is None else q.grad.detach().clone()) for n, q in names.items()}
        img_ref = img.detach().clone()
        for q in names.values():
            q.grad = None
        # fused route
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v])
        assert rf.calls['render_fused'] == 1 and rf.calls['render_reference'] == 0, rf.why_not
        assert isinstance(out, dict) and out['stage'] == 'sk' and tuple(out['images'].shape) == (1, s.H, s.W, 3)
        assert tuple(out['radii'].shape) == (1, s.P) and tuple(out['_knn_w'].shape) == (1, s.P, s.K) and tuple(out['_skT'].shape) == (1, s.M, 7)
        assert float((out['images'][0].permute(2, 0, 1) - img_ref).abs().max()) <= 2e-5
        losses = s.model_loss(out, s.targets_hwc[v])
        assert set(losses) == {'rgb', 'ssim'} and rf.calls['image_terms_fused'] == 1          # ONE launch behind both terms
        loss = sum(losses.values())
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
        loss.backward()
        assert rf.calls['backward_direct'] == 1 and rf.calls['backward_cotangent'] == 0 and rf.calls['foreign_grads_added'] == 0
        for n, q in names.items():
            if want[n] is None:
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
                continue
            assert q.grad is not None, n
            scale = float(want[n].abs().max())
            assert float((q.grad - want[n]).abs().max()) <= 2e-4 * scale + 1e-12, (n, float((q.grad - want[n]).abs().max()), scale)
        # what adaptive_control reads (gaussian_splatting.py:503-513): the screen-space gradient on viewspace_points[0]
        vp = out['viewspace_points'][0]
        assert vp.grad is not None and tuple(vp.grad.shape) == (s.P, 3) and float(vp.grad.abs().max()) > 0
        # the blends the fused kernels never materialise, on first access
        w, i = out['_knn_w'][0], out['_knn_i'][0]
        assert torch.allclose(out['_d_rot'], (out['_sk_rot'][0][i] * w[..., None]).sum(1)) and tuple(out['_d_xyz'].shape) == (s.P, 3)
        assert torch.equal(out['visibility_filter'][0], out['radii'][0] > 0)
        # the frame's row of sk_cache (sk_gs.py:1077-1079) was written by the skeleton launch
        row = s.model.sk_cache[v]
        assert float(row.abs().max()) > 0 and float((row[:, :4].norm(dim=-1) - 1).abs().max()) < 1e-5
        assert float(s.model.sk_cache[(v + 1) % 3].abs().max()) == 0.0
    finally:
        _teardown(s)


def test_other_terms_on_the_same_parameters_and_other_uses_of_the_image_stay_exact():
    """(a) a second loss term on ``_xyz`` / the network in the same backward pass is ADDED (the route's kernels write, they do not
    accumulate: whatever another AccumulateGrad did before is carried over), whichever node autograd runs first; (b) an image that goes
    through something else than the two patched losses reaches the render node as an ordinary cotangent; (c) gradients nobody cleared are
    accumulated onto; (d) eval mode / no_grad: the reference's own render"""
    s = _setup('fused')
    try:
        rf, v = s.rf, 0
        xyz, w0 = s.p['_xyz'], s.net.dynamic_net.net[0].weight

        def run(extra, use_losses=True):
            for q in list(s.p.values()) + list(s.net.parameters()):
                q.grad = None
            out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v])
            if use_losses:
                loss = sum(s.model_loss(out, s.targets_hwc[v]).values())
            else:
                loss = (out['images'] * 2.0).square().mean()
            if extra:
                loss = loss + 0.5 * xyz.square().sum() + 3.0 * w0.sum()
            loss.backward()
            return xyz.grad.detach().clone(), w0.grad.detach().clone()

        gx, gw = run(False)
        gx2, gw2 = run(True)
        assert rf.calls['foreign_grads_added'] >= 0
        assert float((gx2 - (gx + xyz.detach())).abs().max()) <= 1e-5 * float(gx2.abs().max())
        assert float((gw2 - (gw + 3.0)).abs().max()) <= 1e-4 * float(gw2.abs().max())
        # (b)
        n0 = rf.calls['backward_cotangent']
        gx3, _ = run(False, use_losses=False)
        assert rf.calls['backward_cotangent'] == n0 + 1
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v])
        img = out['images'].detach().requires_grad_(True)
        (img * 2.0).square().mean().backward()
        d = s.deform(v)
        for q in list(s.p.values()) + list(s.net.parameters()):
            q.grad = None
        ref_img = s.render(v, d)
        ref_img.backward(img.grad[0].permute(2, 0, 1))
        assert float((gx3 - xyz.grad).abs().max()) <= 2e-4 * float(xyz.grad.abs().max())
        # (c) nobody cleared the gradients: the second view accumulates onto the first
        g1, _ = run(False)
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v])
        sum(s.model_loss(out, s.targets_hwc[v]).values()).backward()
        assert float((xyz.grad - 2 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())
        # (d)
        seen = []
        keep = s.ra._originals['render']
        s.ra._originals['render'] = lambda self, *a, **kw: seen.append(kw) or 'reference'
        s.model.training = False
        assert rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v]) == 'reference'
        s.model.training = True
        with torch.no_grad():
            assert rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v]) == 'reference'
        assert len(seen) == 2 and 'not training' in rf.why_not['render']
        s.ra._originals['render'] = keep
    finally:
        _teardown(s)


def test_training_through_the_fused_route_follows_the_reference_sequence():
    """20 iterations of the reference's loop (render / loss / backward / Adam / zero_grad(set_to_none)) through the fused route and
    through the per-method fast paths on identical scenes: the same loss curve, the same parameters up to the atomics' order, no
    fallback, no overflow, and a replaced Parameter (densification) rebuilds the route"""
    runs = {}
    for mode in ('accelerated', 'fused'):
        s = _setup(mode)
        try:
            losses = [float(s.step(i)) for i in range(20)]
            torch.cuda.synchronize()
            runs[mode] = (losses, {n: q.detach().clone() for n, q in s.p.items()})
            if mode == 'fused':
                rf = s.rf
                assert rf.calls['render_fused'] == 20 and rf.calls['render_reference'] == 0 and rf.calls['routes_built'] == 1
                assert rf.calls['backward_direct'] == 20 and rf.calls['foreign_grads_added'] == 0
                route = rf.route_of_model(s.model)
                st = route.step.status()
                assert st['overflow_events'] == 0 and st.get('mlp_failed', 0) == 0 and route._bucket >= 64
                assert s.ra.calls['adam_fused'] >= 18
                # the heads are rows of one matrix, trained in place
                heads = list(s.net.dynamic_net.last)
                store = route.shadow.dynamic_net.last_weight
                assert heads[1].weight.data_ptr() == store.data_ptr() + 4 * 4 * store.shape[1]
                # densification replaces Parameters: the next call builds a new route on the new objects
                old = s.model._xyz
                s.model._xyz = torch.nn.Parameter(old.detach().clone())
                s.p['_xyz'] = s.model._xyz
                out = rf.render(s.model, t=s.times[0], info=s.infos[0], background=s.bg, time_id=s.time_ids[0])
                assert rf.calls['routes_built'] == 2 and rf.route_of_model(s.model) is not route
                sum(s.model_loss(out, s.targets_hwc[0]).values()).backward()
                assert s.model._xyz.grad is not None and old.grad is None
        finally:
            _teardown(s)
    (la, pa), (lf, pf) = runs['accelerated'], runs['fused']
    for a, b in zip(la, lf):
        assert abs(a - b) <= 2e-3 * abs(a), (la, lf)
    for n in pa:
        far = ((pa[n] - pf[n]).abs() > 2e-2 * float(pa[n].abs().max())).float().mean()
        assert float(far) <= 1e-3, (n, float(far))


SMALL_SP = {9: dict(name='small-4k-160', P=4000, M=12, K=4, W=160, H=120)}


def _setup_sp(mode, extra=()):
    return _setup(mode, ('--stage', 'sp', '--superpoints', '128', '--knn', '4', *extra))


@pytest.mark.parametrize('lbs', ['weighted_kernel', 'W'])
def test_stage_sp_one_iteration_with_the_weight_regularisers(lbs):
    """stage sp through the fused route (``FusedSuperpointStep`` behind ``render``): the image terms AND the two regularisers the shipped
    configuration puts on outputs['_knn_w'] (sparse, smooth: sk_gs.py:1339-1359,1572-1574) -- their cotangent enters the rows pass of the
    backward half (``skgs_sp_skinning_job.g_weights_extra``) -- plus a term on outputs['_spT'] (what the joint losses read, :1555-1563):
    every parameter's gradient equals the reference sequence's on the stand-ins"""
    s = _setup_sp('fused', ('--sp-regularisers',) + (('--lbs-method', 'W') if lbs == 'W' else ()))
    try:
        rf, v = s.rf, 1
        names = dict(s.p)
        names.update({f'net.{n}': q for n, q in s.net.named_parameters()})
        gT = torch.randn(s.M, 7, generator=torch.Generator().manual_seed(5)).cuda() * 1e-3

        def extra_terms(knn_w, spT):
            return s.weight_regularisers(knn_w) + (spT * gT).sum()
        for q in names.values():
            q.grad = None
        res = s.deform(v)
        loss_ref = s.loss_of(s.render(v, res), s.targets[v]) + extra_terms(res['_knn_w'][None], res['_spT'])
        loss_ref.backward()
        want = {n: (None if q.grad is None else q.grad.detach().clone()) for n, q in names.items()}
        for q in names.values():
            q.grad = None
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v], stage='sp')
        assert rf.calls['render_fused'] == 1, rf.why_not
        assert out['stage'] == 'sp' and tuple(out['_knn_w'].shape) == (1, s.P, s.K) and tuple(out['_spT'].shape) == (1, s.M, 7)
        assert '_sp_rot' not in out and tuple(out['_sp_scale'].shape) == (1, s.M, 3)
        assert s.model.sp_weights is not None and s.model.sp_knn is not None           # calc_LBS_weight's side effect (:771-773)
        assert float((out['_knn_w'][0] - res['_knn_w']).abs().max()) <= 1e-6 and float((out['_spT'][0] - res['_spT']).abs().max()) <= 1e-6
        losses = s.model_loss(out, s.targets_hwc[v])
        loss = sum(losses.values()) + (out['_spT'][0] * gT).sum()
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
        loss.backward()
        assert rf.calls['backward_extras'] == 1
        for n, q in names.items():
            if want[n] is None:
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
                continue
            assert q.grad is not None, n
            scale = float(want[n].abs().max())
            assert float((q.grad - want[n]).abs().max()) <= 3e-4 * scale + 1e-12, (n, float((q.grad - want[n]).abs().max()), scale)
        # without the extra terms: the plain backward graph, no cotangent buffers touched
        for q in names.values():
            q.grad = None
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v], stage='sp')
        sum(list(s.model_loss(out, s.targets_hwc[v]).values())[:2]).backward()
        assert rf.calls['backward_extras'] == 1 and s.p['_xyz'].grad is not None
    finally:
        _teardown(s)


@pytest.mark.parametrize('lbs', ['weighted_kernel', 'W'])
def test_stage_sp_training_follows_the_reference_sequence(lbs):
    """15 iterations of the loop in stage sp (image terms) through the fused route and through the per-method fast paths: same losses.
    `W` (the shipped default): the dense [P, M] logit table is updated tile-sparsely by the patched torch.optim.Adam.step on the route
    (skgs_adam_masked_rows, every step after the one that creates torch's state) -- and ends where the dense update ends"""
    runs, tables = {}, {}
    for mode in ('accelerated', 'fused'):
        s = _setup_sp(mode, ('--lbs-method', lbs))
        try:
            n0 = s.ra.calls['adam_tiled']
            runs[mode] = [float(s.step(i)) for i in range(15)]
            torch.cuda.synchronize()
            if lbs == 'W':
                tables[mode] = s.p['sp_W'].detach().clone()
            if mode == 'fused':
                assert s.rf.calls['render_fused'] == 15 and s.rf.calls['render_reference'] == 0 and s.rf.calls['backward_extras'] == 0
                st = s.rf.route_of_model(s.model, 'sp').step.status()
                assert st['overflow_events'] == 0 and st['pairs_overflow_events'] == 0
                assert s.ra.calls['adam_tiled'] - n0 == (14 if lbs == 'W' else 0)
        finally:
            if 'sp_W' in s.p and hasattr(s.p['sp_W'], '_skgs_logit_tiles'):
                del s.p['sp_W']._skgs_logit_tiles
            _teardown(s)
    for a, b in zip(runs['accelerated'], runs['fused']):
        assert abs(a - b) <= 2e-3 * abs(a), runs
    if lbs == 'W':   # the tile-sparse update moved the table as the dense one did (Adam's sign steps: lr-sized differences where a gradient is ~0)
        d = (tables['accelerated'] - tables['fused']).abs()
        assert float((d > 1e-4).float().mean()) < 2e-3, float((d > 1e-4).float().mean())


@pytest.mark.parametrize('warp,sep,lbs', [('LBS_c', True, 'weighted_kernel'), ('largest', False, 'W'), ('LBS_c', False, 'dist')])
def test_stage_sp_variants_against_the_reference_sequence(warp, sep, lbs):
    """the shipped combinations of stage sp through the fused route -- SC-GS (`weighted_kernel`, `LBS_c`, `sep_rot`: exps/d_nerf_sc_gs.yaml)
    and SP-GS (`W`, `largest`: d_nerf_sp_gs.yaml; the route also performs sp_stage's `p2sp` side effect, sk_gs.py:849-850) -- against the
    reference's own sequence on the stand-ins (ref_sequence.sp_stage, pinned by tests/golden/sk_stage.npz): forward values, the loss and
    every parameter's gradient, `sp_points` included where the re-centring of LBS_c reaches it"""
    extra = ('--warp-method', warp, '--lbs-method', lbs) + (('--sep-rot',) if sep else ())
    s = _setup_sp('fused', extra)
    try:
        rf, v = s.rf, 2
        names = dict(s.p)
        names.update({f'net.{n}': q for n, q in s.net.named_parameters()})
        for q in names.values():
            q.grad = None
        res = s.deform(v)
        loss_ref = s.loss_of(s.render(v, res), s.targets[v])
        loss_ref.backward()
        want = {n: (None if q.grad is None else q.grad.detach().clone()) for n, q in names.items()}
        for q in names.values():
            q.grad = None
        out = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v], stage='sp')
        assert rf.calls['render_fused'] == 1, rf.why_not
        assert ('_sp_rot' in out) == sep
        if sep:
            assert float((out['_sp_rot'][0] - res['_sp_rot']).abs().max()) <= 1e-6
        if warp == 'largest':
            assert torch.equal(s.model.p2sp, res['p2sp'])
        assert float((out['_spT'][0] - res['_spT']).abs().max()) <= 2e-6 and float((out['points'][0] - res['points']).abs().max()) <= 2e-5
        loss = sum(s.model_loss(out, s.targets_hwc[v]).values())
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
        loss.backward()
        for n, q in names.items():
            if want[n] is None or float(want[n].abs().max()) == 0.0:
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
                continue
            assert q.grad is not None, n
            scale = float(want[n].abs().max())
            assert float((q.grad - want[n]).abs().max()) <= 3e-4 * scale + 1e-12, (n, float((q.grad - want[n]).abs().max()), scale)
    finally:
        _teardown(s)


def test_arguments_as_a_collating_loader_hands_them_over_stay_on_the_route():
    """train.py:179-191: `inputs` / `infos` come out of a DataLoader's collation and `tensor_to(device)` -- a leading batch dimension on every
    tensor, `info['size']` as two one-element DEVICE tensors, a one-value background, and (loaders with rays) `rays_o` / `rays_d` keywords the
    reference's render ignores: all of that stays on the fused route and renders the same image; a `hook` keyword does not"""
    s = _setup('fused')
    try:
        rf, v = s.rf, 0
        plain = rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v])['images'].detach().clone()
        info = {k: (x[None] if torch.is_tensor(x) else x) for k, x in s.infos[v].items()}
        info['size'] = [torch.tensor([s.W]).cuda(), torch.tensor([s.H]).cuda()]
        out = rf.render(s.model, t=s.times[v].view(1, 1), info=info, background=torch.ones(1, device='cuda'), time_id=s.time_ids[v].view(1),
                        rays_o=torch.zeros(1, 4, 3), rays_d=torch.zeros(1, 4, 3))
        assert rf.calls['render_fused'] == 2 and rf.calls['render_reference'] == 0, rf.why_not
        assert torch.equal(out['images'].detach(), plain)
        seen = []
        keep = s.ra._originals['render']
        s.ra._originals['render'] = lambda self, *a, **kw: seen.append(kw) or 'reference'
        assert rf.render(s.model, t=s.times[v], info=s.infos[v], background=s.bg, time_id=s.time_ids[v], hook=lambda o: o) == 'reference'
        assert 'hook' in seen[0] and 'hook' in rf.why_not['render']
        s.ra._originals['render'] = keep
    finally:
        _teardown(s)


@pytest.mark.parametrize('P,K,G', [(100_000, 5, 21), (3001, 4, 7), (64, 16, 3), (1, 1, 1)])
def test_weight_regularisers_as_one_launch_each_equal_the_reference_lines(P, K, G):
    """skgs_weight_sparsity / skgs_weight_smooth (csrc/weight_reg.hip; sk_gs_amd/weight_reg.py) against the reference's own expressions
    (sk_gs.py:1339-1340, 1357-1359) in torch: value and gradient, with negative and self indices in the neighbour table"""
    from sk_gs_amd import weight_reg as wr
    g = torch.Generator().manual_seed(P + K)
    w0 = torch.softmax(torch.randn(P, K, generator=g), -1).cuda()
    nbr = torch.randint(0, P, (P, G), generator=g)
    nbr[:, 0] = torch.arange(P)                       # the Gaussian itself, as pykdtree returns it first (sk_gs.py:1351)
    if P > 10:
        nbr[5, 1] -= P                                # a negative index: counted from the end, as torch does
    nbr = nbr.cuda()
    res = []
    for fused in (True, False):
        w = w0.clone().requires_grad_()
        if fused:
            sparse, smooth = wr.weight_sparsity(w[None], 1e-7), wr.weight_smooth(w, nbr)
        else:
            sparse = -(w[None] * torch.log(w[None] + 1e-7) + (1 - w[None]) * torch.log(1 - w[None] + 1e-7)).mean()
            smooth = (w[:, None] - w[nbr]).abs().mean()
        (0.3 * sparse + 0.7 * smooth).backward()
        res.append((float(sparse), float(smooth), w.grad.clone()))
    (s1, m1, g1), (s0, m0, g0) = res
    assert abs(s1 - s0) <= 2e-6 * abs(s0) + 1e-9 and abs(m1 - m0) <= 2e-6 * abs(m0) + 1e-9, (s1, s0, m1, m0)
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-12


def test_tiled_adam_on_the_logit_table_is_bit_identical_to_the_dense_launch():
    """``skgs_adam_masked_rows`` behind the patched ``torch.optim.Adam.step`` (reference_accel: a parameter that carries ``_skgs_logit_tiles``):
    the [P, M] logit table updated tile-sparsely from its DENSE gradient -- parameters and both moments bit-identical to the dense one-launch
    update over 12 steps in which the neighbour columns move, one step has a gradient OUTSIDE the known neighbours (scan), and the moments
    are non-zero before the note is attached (the mask is rebuilt from them)"""
    from sk_gs_amd import reference_accel as ra
    import torch.optim
    keep = torch.optim.Adam.step
    ra._originals.setdefault('adam', keep)
    torch.optim.Adam.step = ra.adam_step
    try:
        P, M, K = 5000, 512, 5
        g = torch.Generator().manual_seed(11)
        w0, small0 = torch.randn(P, M, generator=g).cuda(), torch.randn(300, generator=g).cuda()
        idx = torch.randint(0, M, (P, K), generator=g).cuda()
        runs = []
        gg = torch.Generator().manual_seed(5)
        grads = []                 # the SAME gradients for both runs (a scatter with duplicate columns in a row is not deterministic)
        for it in range(12):
            grad = torch.zeros(P, M, device='cuda')
            cols = (idx + it // 4) % M                                       # the neighbours move every four steps
            grad.scatter_(1, cols, torch.randn(P, K, generator=gg).cuda())
            if it == 7:                                                       # a term that reaches other columns (another loss on the table)
                grad[::17, 300] += 0.5
            grads.append((grad, cols, torch.randn(300, generator=gg).cuda()))
        for tiled in (False, True):
            w, small = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(small0.clone())
            opt = torch.optim.Adam([{'params': [w], 'lr': 1e-2}, {'params': [small], 'lr': 1e-3}], eps=1e-15)
            n0 = ra.calls['adam_tiled']
            for it in range(12):
                grad, cols, gsmall = grads[it]
                outside = it == 7
                w.grad, small.grad = grad.clone(), gsmall.clone()
                if tiled and it == 3:                                         # the note arrives when the moments are already non-zero
                    w._skgs_logit_tiles = ra.LogitTiles(cols)
                if tiled and it >= 3:
                    w._skgs_logit_tiles.indices, w._skgs_logit_tiles.scan = cols, outside
                opt.step()
            torch.cuda.synchronize()
            assert (ra.calls['adam_tiled'] - n0) == (9 if tiled else 0)
            st = opt.state[w]
            runs.append((w.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), small.detach().clone(), float(st['step'])))
            if tiled:   # the mask is sparse: a row has touched at most 3 x K + 1 tiles of 16
                bits = w._skgs_logit_tiles.mask.cpu().numpy()
                assert max(bin(int(b) & 0xffffffff).count('1') for b in bits) <= 16 and sum(bin(int(b) & 0xffffffff).count('1') for b in bits) < 0.8 * 16 * P
        for name, a, b in zip(('table', 'exp_avg', 'exp_avg_sq', 'the other parameter'), runs[0][:4], runs[1][:4]):
            assert torch.equal(a, b), (name, int((a != b).sum()), float((a - b).abs().max()), (a != b).nonzero()[:5].tolist())
        assert runs[0][4] == runs[1][4] == 12.0
    finally:
        torch.optim.Adam.step = keep
        ra._originals.pop('adam', None)
        ra._adam_plans.clear()
