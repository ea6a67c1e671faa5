"""OverflowGuard (sk_gs_amd/overflow.py): a forward whose tile lists exceed the binning capacity drops splats and raises a
sticky device flag; the guard notices at its next check, rolls the training state back to its last snapshot and the caller
redoes those iterations with a larger capacity -- the run ends where a run with enough capacity from the start ends."""
import pytest
import torch

from helpers import assert_close_robust

pytestmark = pytest.mark.gpu


def _train(capacity, iters, guard_every, views, targets, seed=3):
    from sk_gs_amd import scene
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.model import SkinnedGaussians
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.overflow import OverflowGuard
    from sk_gs_amd.train_step import GraphedSteps
    dev = torch.device('cuda')
    P, M, K, W, H = 5000, 8, 4, 128, 96
    model = SkinnedGaussians(P, M, K, num_frames=len(views), seed=seed, scale_mult=2.0, deform_net=True).to(dev)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt = FusedAdam(model.param_groups(lr=2e-3), eps=1e-15)
    step = FusedViewStep(model, W, H, capacity=capacity, densify_stats=True)
    step.forward_backward(views[0], 0, targets[0])  # warm-up outside any capture (may already overflow: counter restarts)
    step.geom[:256].zero_()
    graphs = GraphedSteps(lambda v: (step.forward_backward(views[v], v, targets[v]), opt.step()))
    guard = OverflowGuard(step, opt, every=guard_every) if guard_every else None
    it, log = 0, []
    while it < iters:
        graphs(it % len(views))
        act = guard.after_step(it) if guard else None
        if act is not None:
            log.append((it, act))
            step.grow_capacity(2.0)
            graphs.graphs.clear()
            guard.rebind(step)
            it = act[1]
            continue
        it += 1
    torch.cuda.synchronize()
    return model, opt, step, log


def test_overflow_is_detected_rolled_back_and_redone():
    from sk_gs_amd import _C, scene
    from sk_gs_amd.model import SkinnedGaussians
    dev = torch.device('cuda')
    W, H, V = 128, 96, 4
    views = [scene.raster_settings_from_camera(scene.make_camera(W, H, seed=20 + v), sh_degree=3, colmap=True, device=dev)
             for v in range(V)]
    probe = SkinnedGaussians(5000, 8, 4, num_frames=V, seed=3, scale_mult=2.0, deform_net=True).to(dev)
    _C.config.sync_num_rendered = True
    with torch.no_grad():
        outs = [probe.render(views[v], time_id=v) for v in range(V)]
    Rs = [o['buffer'].R for o in outs]
    targets = [(o['images'] * 0.7 + 0.1).detach().contiguous() for o in outs]
    assert max(Rs) > min(Rs)
    tight = (max(Rs) + sorted(Rs)[-2]) // 2 if sorted(Rs)[-2] < max(Rs) else max(Rs) - 1  # fits all views but the largest
    iters = 12
    ref_model, ref_opt, ref_step, ref_log = _train(4 * max(Rs), iters, 0, views, targets)
    assert ref_step.status()['overflow_events'] == 0
    model, opt, step, log = _train(tight, iters, 4, views, targets)
    assert len(log) >= 1 and log[0][1][0] == 'redo' and log[0][1][1] == 0, log   # first check (after iteration 3) rolls back to 0
    assert step.status()['overflow_events'] == 0                                   # the redone run fits
    assert float(opt.step_count) == float(ref_opt.step_count) == iters              # every iteration counted exactly once
    for (n, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
        # twelve Adam steps amplify the float-atomics ordering noise of the two runs (the first Adam steps are sign-like)
        assert_close_robust(p.data, q.data, 1e-3, 1e-3, name=f'{n} after the redo')
    assert torch.equal(step.denom, ref_step.denom)
