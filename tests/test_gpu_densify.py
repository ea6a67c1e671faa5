"""FusedAdam.change_optimizer (the optimizer-state surgery of densification, gaussian_splatting.py:515-563) against
torch.optim.Adam manipulated the same way, and the clone / split / prune operations of sk_gs_amd.densify."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _torch_surgery(opt, group, new_param, new_state):
    old = group['params'][0]
    opt.state.pop(old)
    group['params'][0] = new_param
    opt.state[new_param] = new_state


def test_change_optimizer_matches_torch_adam_surgery():
    from sk_gs_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    a0, b0 = torch.randn(50, 3, generator=g).cuda(), torch.randn(50, 4, generator=g).cuda()
    pa, pb = torch.nn.Parameter(a0.clone()), torch.nn.Parameter(b0.clone())
    ta, tb = torch.nn.Parameter(a0.clone()), torch.nn.Parameter(b0.clone())
    fused = FusedAdam([{'params': [pa], 'lr': 0.01, 'name': 'a'}, {'params': [pb], 'lr': 0.02, 'name': 'b'}], eps=1e-15)
    ref = torch.optim.Adam([{'params': [ta], 'lr': 0.01, 'name': 'a'}, {'params': [tb], 'lr': 0.02, 'name': 'b'}], eps=1e-15)

    def step(params_f, params_t):
        for pf, pt in zip(params_f, params_t):
            gr = torch.randn(pf.shape, generator=g).cuda()
            pf.grad, pt.grad = gr.clone(), gr.clone()
        fused.step()
        ref.step()

    for _ in range(3):
        step([pa, pb], [ta, tb])
    # ---- prune rows of both groups
    keep = (torch.rand(50, generator=g) > 0.3).cuda()
    new = fused.change_optimizer(keep, ['a', 'b'], op='prune')
    pa, pb = new['a'], new['b']
    for grp, old in zip(ref.param_groups, (ta, tb)):
        st = ref.state[old]
        p = torch.nn.Parameter(old.data[keep].clone())
        _torch_surgery(ref, grp, p, {'step': st['step'], 'exp_avg': st['exp_avg'][keep].clone(),
                                     'exp_avg_sq': st['exp_avg_sq'][keep].clone()})
    ta, tb = ref.param_groups[0]['params'][0], ref.param_groups[1]['params'][0]
    assert pa.shape == ta.shape and pa.shape[0] == int(keep.sum())
    step([pa, pb], [ta, tb])
    assert rel_err(pa, ta) <= 1e-6 and rel_err(pb, tb) <= 1e-6
    # ---- concat new rows to group a, replace group b
    extra = torch.randn(7, 3, generator=g).cuda()
    newb = torch.randn(pb.shape, generator=g).cuda()
    pa = fused.change_optimizer({'a': extra}, op='concat')['a']
    pb = fused.change_optimizer(newb, name='b', op='replace')['b']
    st = ref.state[ta]
    p = torch.nn.Parameter(torch.cat([ta.data, extra]))
    _torch_surgery(ref, ref.param_groups[0], p, {'step': st['step'], 'exp_avg': torch.cat([st['exp_avg'], torch.zeros_like(extra)]),
                                                 'exp_avg_sq': torch.cat([st['exp_avg_sq'], torch.zeros_like(extra)])})
    st = ref.state[tb]
    p = torch.nn.Parameter(newb.clone())
    _torch_surgery(ref, ref.param_groups[1], p, {'step': st['step'], 'exp_avg': torch.zeros_like(newb),
                                                 'exp_avg_sq': torch.zeros_like(newb)})
    ta, tb = ref.param_groups[0]['params'][0], ref.param_groups[1]['params'][0]
    for _ in range(2):
        step([pa, pb], [ta, tb])
    assert pa.shape == (int(keep.sum()) + 7, 3)
    assert rel_err(pa, ta) <= 1e-6 and rel_err(pb, tb) <= 1e-6


def test_densify_and_prune_keep_model_optimizer_and_step_consistent():
    """clone + split + prune on a skinned model, then a FusedViewStep on the new size still trains"""
    from sk_gs_amd import _C, densify, scene
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.model import SkinnedGaussians
    from sk_gs_amd.optim import FusedAdam
    dev = torch.device('cuda')
    P, M, K, W, H = 3000, 8, 4, 128, 96
    model = SkinnedGaussians(P, M, K, num_frames=2, seed=5, scale_mult=2.0).to(dev)
    opt = FusedAdam(model.param_groups(lr=1e-3), eps=1e-15)
    stats = densify.DensifyStats(P, dev)
    gen = torch.Generator(device='cuda').manual_seed(1)
    stats.xyz_gradient_accum = torch.rand(P, 1, device=dev, generator=gen) * 4e-4
    stats.denom = torch.ones(P, 1, device=dev)
    stats.max_radii2D = torch.rand(P, device=dev, generator=gen) * 30
    extent = 5.0
    grads = (stats.xyz_gradient_accum / stats.denom)
    n_big = int(((grads.squeeze() >= 2e-4) & (torch.exp(model._scaling).amax(1) > 0.01 * extent)).sum())
    n_small = int(((grads.squeeze() >= 2e-4) & (torch.exp(model._scaling).amax(1) <= 0.01 * extent)).sum())
    densify.densify(model, opt, stats, max_grad=2e-4, extent=extent, generator=gen)
    # clone adds n_small; split (on the grown set, gradients zero-padded) adds 2 * n_big and removes n_big
    assert model.P == P + n_small + n_big
    for attr, name in densify._names(model).items():  # (the per-Gaussian tensors this model has: no hyper features in stage sk)
        p = getattr(model, attr)
        assert p.shape[0] == model.P and isinstance(p, torch.nn.Parameter)
        st = opt.state[p]
        assert st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape
    assert stats.denom.shape == (model.P, 1) and float(stats.denom.abs().max()) == 0.0
    stats.max_radii2D = torch.rand(model.P, device=dev, generator=gen) * 30
    before = model.P
    densify.prune(model, opt, stats, min_opacity=0.05, extent=extent, max_screen_size=20.0)
    assert 0 < model.P < before and stats.max_radii2D.shape == (model.P,)
    densify.reset_opacity(model, opt)
    assert float(torch.sigmoid(model._opacity).max()) <= 0.0100001
    # the step still runs on the new size and lowers its loss
    cam = scene.make_camera(W, H, seed=0)
    rs = scene.raster_settings_from_camera(cam, sh_degree=3, colmap=True, device=dev)
    _C.config.sync_num_rendered = True
    with torch.no_grad():
        out = model.render(rs, time_id=0)
        target = (out['images'] * 0.5 + 0.2).contiguous()
    step = FusedViewStep(model, W, H, capacity=int(out['buffer'].R * 3) + 4096)
    step.forward_backward(rs, 0, target)
    first = float(step.loss3[0])
    for _ in range(40):
        opt.step()
        step.forward_backward(rs, 0, target)
    assert float(step.loss3[0]) < first


def test_density_control_replays_the_reference_run():
    """tests/golden/densify.npz: GaussianSplatting.densify_and_clone / prune / reset_opacity / densify_and_split of the
    reference (gaussian_splatting.py:515-655) run on CPU over torch.optim.Adam, with Adam steps in between.  The same
    sequence through sk_gs_amd.densify + FusedAdam must give the same parameters, moments and statistics."""
    import os
    import numpy as np
    from sk_gs_amd import densify
    from sk_gs_amd.optim import FusedAdam
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'densify.npz'))
    names = ['_xyz', '_features_dc', '_features_rest', '_opacity', '_scaling', '_rotation']
    dev = torch.device('cuda')

    class Gaussians(torch.nn.Module):  # the six parameters of the reference's module, nothing else
        def __init__(self):
            super().__init__()
            for n in names:
                setattr(self, n, torch.nn.Parameter(torch.tensor(z['init.' + n], device=dev)))
            self.P = self._xyz.shape[0]

    m = Gaussians()
    opt = FusedAdam([{'params': [getattr(m, n)], 'lr': float(lr), 'name': densify.PARAM_NAMES_MAP[n]}
                     for n, lr in zip(names, z['lr'])], eps=1e-15)
    stats = densify.DensifyStats(m.P, dev)

    def adam_step(tag):
        for n in names:
            getattr(m, n).grad = torch.tensor(z[f'{tag}.grad.{n}'], device=dev)
        opt.step()

    def check(tag, skip=(), tol=2e-6):
        for n in names:
            p = getattr(m, n)
            assert tuple(p.shape) == z[f'{tag}.{n}'].shape, (tag, n, tuple(p.shape))
            if n in skip:
                continue
            st = opt.state[p]
            assert rel_err(p, z[f'{tag}.{n}']) <= tol, (tag, n)
            assert rel_err(st['exp_avg'], z[f'{tag}.m.{n}']) <= tol and rel_err(st['exp_avg_sq'], z[f'{tag}.v.{n}']) <= tol, (tag, n)

    def set_stats(tag):
        stats.xyz_gradient_accum = torch.tensor(z[f'{tag}.in_accum'], device=dev)
        stats.denom = torch.tensor(z[f'{tag}.in_denom'], device=dev)
        stats.max_radii2D = torch.tensor(z[f'{tag}.in_radii'], device=dev)

    def grads_of_stats():
        g = stats.xyz_gradient_accum / stats.denom
        g[g.isnan()] = 0.0
        return g

    extent, thr = 5.0, 2e-4
    adam_step('s0'), adam_step('s1')
    check('after_steps')
    set_stats('clone')
    densify.densify_and_clone(m, opt, grads_of_stats(), thr, 0.01 * extent, stats)
    check('after_clone')
    assert float(stats.denom.abs().max()) == 0.0 and stats.denom.shape == z['after_clone.denom'].shape
    set_stats('prune')
    densify.prune(m, opt, stats, min_opacity=0.05, extent=extent, max_screen_size=20.0)
    check('after_prune')
    assert np.array_equal(stats.max_radii2D.cpu().numpy(), z['after_prune.radii'])
    adam_step('s2')
    check('after_step2')
    densify.reset_opacity(m, opt)
    check('after_reset')
    set_stats('split')
    n_sel = int(z['split.selected'].sum())
    before = {n: getattr(m, n).detach().clone() for n in names}
    densify.densify_and_split(m, opt, grads_of_stats(), thr, 0.01 * extent, N=2, stats=stats,
                              generator=torch.Generator(device='cuda').manual_seed(3))
    # the new positions are random samples (another generator than the reference's CPU one): everything else must agree
    check('after_split', skip=('_xyz',))
    kept = (~torch.tensor(z['split.selected'], device=dev))
    n_kept = int(kept.sum())
    assert torch.equal(m._xyz[:n_kept], before['_xyz'][kept])
    assert rel_err(m._xyz[:n_kept], z['after_split._xyz'][:n_kept]) <= 2e-6
    # a sample lies within a few standard deviations of its parent: |R^T (x_new - mu)| / sigma is a standard normal draw
    sel = ~kept
    mu = before['_xyz'][sel].repeat(2, 1)
    R = densify.quaternion_to_R(before['_rotation'][sel]).repeat(2, 1, 1)
    sig = torch.exp(before['_scaling'][sel]).repeat(2, 1)
    zed = torch.bmm(R.transpose(1, 2), (m._xyz[n_kept:] - mu)[..., None]).squeeze(-1) / sig
    assert zed.shape == (2 * n_sel, 3) and float(zed.abs().max()) < 6.0 and 0.5 < float(zed.std()) < 1.5
    # the state of the appended rows starts from zero and the next step works on the new shapes
    for n in names:
        if n != '_xyz':
            getattr(m, n).grad = torch.tensor(z[f's3.grad.{n}'], device=dev)
    m._xyz.grad = torch.tensor(z['s3.grad._xyz'], device=dev)
    opt.step()
    check('after_step3', skip=('_xyz',))
