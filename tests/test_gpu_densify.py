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
    for attr, name in densify.PARAM_NAMES_MAP.items():
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
