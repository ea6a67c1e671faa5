"""A captured training step that survives densification (sk_gs_amd/capacity.py).

With ``model.enable_capacity(P_cap)`` the per-Gaussian parameters, their gradient slots, Adam moments and the step's
workspaces hold ``P_cap`` rows; the kernels are launched for the capacity and read the live count from a device word; clone /
split / prune (networks/gaussian_splatting.py:565-636) rewrite the rows IN PLACE.  Checked here:

  * a capacity-mode step computes what the plain step computes (image bit for bit, gradients to the atomics' order);
  * in-place clone + split + prune leave exactly the parameters, moments and statistics of the reference-shaped surgery
    (``FusedAdam.gather_rows`` into new tensors), bit for bit;
  * ONE hipGraph captured before a densification keeps replaying after it -- no re-capture -- and its next step renders and
    differentiates the densified model exactly like a runtime built from scratch for it;
  * growing beyond the capacity raises ``CapacityExceeded`` and leaves the state untouched.
"""
import copy

import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

P, M, K, W, H, V = 6000, 12, 4, 128, 96, 3


def _scene(dev):
    from sk_gs_amd import scene
    settings = [scene.raster_settings_from_camera(scene.make_camera(W, H, seed=30 + v), sh_degree=3, colmap=True, device=dev)
                for v in range(V)]
    targets = torch.rand(V, 3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    return settings, targets


def _model(dev, seed=5):
    from sk_gs_amd.model import SkinnedGaussians
    return SkinnedGaussians(P, M, K, sh_degree=3, num_frames=V, seed=seed, scale_mult=2.0, deform_net=True,
                            learn_joints=True).to(dev)


def _runtime(model, settings, targets, lr=1e-3, graphed=True):
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.train_step import FusedTrainStep, GraphedSteps
    from sk_gs_amd.view_parallel import FlatGradBuffer
    from sk_gs_amd.view_slot import ViewTable
    dev = targets.device
    table = ViewTable(settings, [float(model.frame_times[v]) for v in range(V)], list(range(V)), targets, dev)
    buf = FlatGradBuffer(model.parameters())
    step = FusedViewStep(model, W, H, capacity=2_000_000, densify_stats=True, view_table=table)
    opt = FusedAdam(model.param_groups(lr=lr), eps=1e-15)
    train = FusedTrainStep(step, opt)
    assert train.fused
    table.set_order(list(range(V)))
    run = GraphedSteps(lambda _: train(), collect_garbage=False) if graphed else (lambda _: train())
    return table, buf, step, opt, train, run


def _fill_stats(step, n, seed):
    gen = torch.Generator(device='cuda').manual_seed(seed)
    step.xyz_gradient_accum = torch.rand(n, 1, device='cuda', generator=gen) * 3e-4
    step.denom = torch.ones(n, 1, device='cuda')
    step.max_radii2D = torch.rand(n, device='cuda', generator=gen) * 40


def test_capacity_step_equals_the_plain_step():
    dev = torch.device('cuda')
    settings, targets = _scene(dev)
    plain = _model(dev)
    capm = copy.deepcopy(plain)
    capm.enable_capacity(int(P * 1.5))
    assert capm._xyz.shape == plain._xyz.shape and capm._xyz.is_contiguous() and capm.capacity.P_cap == 9000
    outs = []
    for m in (plain, capm):
        table, buf, step, opt, train, run = _runtime(m, settings, targets, graphed=False)
        table.select(1)
        step.forward_backward()
        torch.cuda.synchronize()
        st = step.status()
        assert st['overflow_events'] == 0 and st['mlp_failed'] == 0
        outs.append((step.image.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}, step.radii[:P].clone(),
                     step.denom.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    for n in outs[0][1]:
        assert outs[0][1][n].shape == outs[1][1][n].shape
        assert rel_err(outs[1][1][n], outs[0][1][n]) <= 2e-5, n
    # the rows behind the live count are dead: radius 0, no statistics, no tile instances
    assert int(step.radii[P:].abs().max()) == 0 and float(step._den_store[P:].abs().max()) == 0.0


def test_in_place_surgery_equals_the_reference_shaped_surgery():
    from sk_gs_amd import densify
    from sk_gs_amd.optim import CapacityExceeded, FusedAdam
    dev = torch.device('cuda')
    settings, targets = _scene(dev)
    plain = _model(dev)
    capm = copy.deepcopy(plain)
    capm.enable_capacity(int(P * 1.6))
    rt = {}
    for name, m in (('plain', plain), ('cap', capm)):
        table, buf, step, opt, train, run = _runtime(m, settings, targets, graphed=False)
        for _ in range(2):  # two real steps: non-trivial moments
            run(0)
        rt[name] = (m, step, opt, buf)
    torch.cuda.synchronize()
    # the two replicas trained alike up to the atomics' order: make the second an exact copy of the first
    (m0, s0, o0, _), (m1, s1, o1, _) = rt['plain'], rt['cap']
    with torch.no_grad():
        for (n, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
            b.copy_(a)
            o1.state[b]['exp_avg'].copy_(o0.state[a]['exp_avg']), o1.state[b]['exp_avg_sq'].copy_(o0.state[a]['exp_avg_sq'])
    ptrs = {n: (p.data_ptr(), p.grad.data_ptr(), o1.state[p]['exp_avg'].data_ptr()) for n, p in m1.named_parameters()}
    params_before = dict(m1.named_parameters())
    for m, s, o in ((m0, s0, o0), (m1, s1, o1)):
        _fill_stats(s, m.P, seed=9)
        densify.densify(m, o, s, max_grad=2e-4, extent=4.0, generator=torch.Generator(device='cuda').manual_seed(77))
        assert m.P > P
        _fill_stats(s, m.P, seed=10)
        densify.prune(m, o, s, min_opacity=0.05, extent=4.0, max_screen_size=30.0)
    torch.cuda.synchronize()
    assert m0.P == m1.P and P * 0.5 < m1.P != P
    assert int(m1.capacity.live.item()) == m1.P
    for (n, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        assert a.shape == b.shape and torch.equal(a, b), n
        assert torch.equal(o0.state[a]['exp_avg'], o1.state[b]['exp_avg']), n
        assert torch.equal(o0.state[a]['exp_avg_sq'], o1.state[b]['exp_avg_sq']), n
        assert b.grad.shape == b.shape
        # in place: the same Parameter objects over the same storage
        assert dict(m1.named_parameters())[n] is params_before[n]
        assert (b.data_ptr(), b.grad.data_ptr(), o1.state[b]['exp_avg'].data_ptr()) == ptrs[n], n
    for name in ('xyz_gradient_accum', 'denom', 'max_radii2D'):
        assert torch.equal(getattr(s0, name), getattr(s1, name)), name
    # more rows than the capacity holds: refused, nothing changed
    before = m1._xyz.clone()
    rows = torch.arange(m1.P, device=dev).repeat(3)
    with pytest.raises(CapacityExceeded):
        o1.gather_rows(['xyz', 'f_dc', 'f_rest', 'opacity', 'scaling', 'rotation', 'sp_W'], rows, m1.P)
    assert torch.equal(m1._xyz, before)


def test_one_captured_graph_keeps_training_through_clone_split_and_prune():
    from sk_gs_amd import densify
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.model import SkinnedGaussians
    dev = torch.device('cuda')
    settings, targets = _scene(dev)
    model = _model(dev)
    model.enable_capacity(2 * P)
    table, buf, step, opt, train, graph = _runtime(model, settings, targets, lr=2e-3)
    for i in range(4):
        graph(0)
    torch.cuda.synchronize()
    assert len(graph.graphs) == 1 and float(opt.step_count.item()) == 4
    sizes = [model.P]
    for event in range(2):
        _fill_stats(step, model.P, seed=20 + event)
        densify.densify(model, opt, step, max_grad=2e-4, extent=4.0, generator=torch.Generator(device='cuda').manual_seed(event))
        _fill_stats(step, model.P, seed=40 + event)
        densify.prune(model, opt, step, min_opacity=0.02, extent=4.0, max_screen_size=35.0)
        sizes.append(model.P)
        step.reset_densify_stats()
        # the next step of the OLD graph, against a runtime built from scratch for the densified model
        view = int(table.cursor[0].item() - 1) % V
        fresh = SkinnedGaussians(model.P, M, K, sh_degree=3, num_frames=V, seed=5, scale_mult=2.0, deform_net=True,
                                 learn_joints=True).to(dev)
        with torch.no_grad():
            for (n, a), (_, b) in zip(model.named_parameters(), fresh.named_parameters()):
                assert a.shape == b.shape, n
                b.copy_(a)
        for p in fresh.parameters():
            p.grad = torch.zeros_like(p)
        fstep = FusedViewStep(fresh, W, H, capacity=2_000_000, densify_stats=True)
        fstep.forward_backward(settings[view], view, targets[view])
        graph(0)
        torch.cuda.synchronize()
        assert len(graph.graphs) == 1, 'the step was re-captured'
        assert step.status()['overflow_events'] == 0 and step.status()['mlp_failed'] == 0
        assert torch.equal(step.image, fstep.image), event
        assert torch.equal(step.radii[:model.P], fstep.radii) and int(step.radii[model.P:].abs().max()) == 0
        assert torch.equal(step.denom, fstep.denom)
        for (n, a), (_, b) in zip(model.named_parameters(), fresh.named_parameters()):
            assert rel_err(a.grad, b.grad) <= 2e-5, (event, n)
        for i in range(3):
            graph(0)
    torch.cuda.synchronize()
    assert sizes[1] != sizes[0] and sizes[2] != sizes[1], sizes
    assert float(opt.step_count.item()) == 4 + 2 * 4 and len(graph.graphs) == 1
    # the model kept learning: the loss of the last step is finite and the parameters are finite
    assert torch.isfinite(step.loss3).all() and all(torch.isfinite(p).all() for p in model.parameters())


def test_growing_the_capacity_keeps_parameters_and_moments():
    """``CapacityExceeded`` -> ``RowCapacity.grow``: parameters and Adam moments move into larger storage unchanged, the
    densification that did not fit then does, and a rebuilt runtime trains on (one rebuild + re-capture per capacity
    doubling instead of one per event)"""
    from sk_gs_amd import densify
    from sk_gs_amd.optim import CapacityExceeded
    dev = torch.device('cuda')
    settings, targets = _scene(dev)
    model = _model(dev)
    model.enable_capacity(int(P * 1.1))
    table, buf, step, opt, train, graph = _runtime(model, settings, targets, lr=2e-3)
    for _ in range(3):
        graph(0)
    torch.cuda.synchronize()
    snap = {n: (p.detach().clone(), opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone())
            for n, p in model.named_parameters()}
    _fill_stats(step, model.P, seed=3)
    with pytest.raises(CapacityExceeded):
        densify.densify(model, opt, step, max_grad=2e-4, extent=4.0, generator=torch.Generator(device='cuda').manual_seed(1))
    assert model.P == P and all(torch.equal(p, snap[n][0]) for n, p in model.named_parameters())
    stats = (step.xyz_gradient_accum.clone(), step.denom.clone(), step.max_radii2D.clone())
    model.capacity.grow(model, 2 * P, optimizer=opt)
    for n, p in model.named_parameters():
        assert torch.equal(p, snap[n][0]) and torch.equal(opt.state[p]['exp_avg'], snap[n][1]), n
        assert torch.equal(opt.state[p]['exp_avg_sq'], snap[n][2]), n
    table, buf, step, _, train, graph = _runtime_with(model, opt, settings, targets)
    step.xyz_gradient_accum, step.denom, step.max_radii2D = stats
    densify.densify(model, opt, step, max_grad=2e-4, extent=4.0, generator=torch.Generator(device='cuda').manual_seed(1))
    assert model.P > P and int(model.capacity.live.item()) == model.P
    step.reset_densify_stats()
    for _ in range(3):
        graph(0)
    torch.cuda.synchronize()
    assert step.status()['overflow_events'] == 0 and float(opt.step_count.item()) == 6
    assert torch.isfinite(step.loss3).all() and int(step.radii[model.P:].abs().max()) == 0


def _runtime_with(model, opt, settings, targets):
    """a rebuilt runtime around an EXISTING optimizer (after RowCapacity.grow)"""
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.train_step import FusedTrainStep, GraphedSteps
    from sk_gs_amd.view_parallel import FlatGradBuffer
    from sk_gs_amd.view_slot import ViewTable
    dev = targets.device
    table = ViewTable(settings, [float(model.frame_times[v]) for v in range(V)], list(range(V)), targets, dev)
    buf = FlatGradBuffer(model.parameters())
    step = FusedViewStep(model, W, H, capacity=2_000_000, densify_stats=True, view_table=table)
    opt.rebind()
    train = FusedTrainStep(step, opt)
    table.set_order(list(range(V)))
    return table, buf, step, opt, train, GraphedSteps(lambda _: train(), collect_garbage=False)
