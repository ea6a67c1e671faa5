"""End to end: does training learn?  A student scene (colours halved, opacities lowered, positions jittered, its own randomly
initialised deform network) is fitted to images rendered from a teacher, through the fused product step and through the
reference-shaped operator path; the loss of the last iterations must be well below that of the first (a wrong sign or a
missing factor in any gradient, or a broken optimizer piece, stalls or diverges instead)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

P, M, K, W, H, V = 6000, 8, 4, 160, 128, 4


def _scene():
    from sk_gs_amd import _C, scene
    from sk_gs_amd.model import SkinnedGaussians
    dev = torch.device('cuda')
    teacher = SkinnedGaussians(P, M, K, num_frames=V, seed=3, deform_net=True, scale_mult=2.0).to(dev)
    student = SkinnedGaussians(P, M, K, num_frames=V, seed=3, deform_net=True, scale_mult=2.0, learn_joints=True).to(dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        student._features_dc.mul_(0.4)
        student._opacity.sub_(0.7)
        student._xyz.add_(0.01 * torch.randn(P, 3, generator=g).to(dev))
        student._scaling.add_(0.1)
    views = [scene.raster_settings_from_camera(scene.make_camera(W, H, seed=i), sh_degree=3, colmap=True, device=dev)
             for i in range(V)]
    bg = torch.ones(3, device=dev)
    _C.config.sync_num_rendered = True
    with torch.no_grad():
        outs = [teacher.render(views[v], time_id=v, background=bg) for v in range(V)]
    targets = torch.stack([o['images'] for o in outs]).contiguous()
    R = max(o['buffer'].R for o in outs)
    return student, views, targets, bg, R


def _psnr(model, views, targets, bg):
    with torch.no_grad():
        mse = sum(float(((model.render(views[v], time_id=v, background=bg)['images'] - targets[v]) ** 2).mean()) for v in range(V)) / V
    return -10.0 * torch.log10(torch.tensor(mse)).item()


def test_fused_training_fits_the_teacher_images():
    from sk_gs_amd.fused_step import FusedViewStep
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.train_step import FusedTrainStep
    from sk_gs_amd.view_parallel import ViewParallel
    model, views, targets, bg, R = _scene()
    before = _psnr(model, views, targets, bg)
    opt = FusedAdam(model.param_groups(lr=2e-3), eps=1e-15)
    ViewParallel(model.parameters())
    step = FusedViewStep(model, W, H, capacity=4 * R + 4096, background=bg)
    opt.rebind()
    train = FusedTrainStep(step, opt)
    losses = []
    for it in range(240):
        v = it % V
        train(views[v], v, targets[v])
        losses.append(step.loss3[0].clone())
    losses = torch.stack(losses).cpu()
    st = step.status()
    assert st['overflow_events'] == 0 and st.get('mlp_failed', 0) == 0
    first, last = float(losses[:8].mean()), float(losses[-8:].mean())
    after = _psnr(model, views, targets, bg)
    print(f'fused: loss {first:.4f} -> {last:.4f}, PSNR {before:.2f} -> {after:.2f} dB')
    assert bool(torch.isfinite(losses).all())
    assert last < 0.2 * first, (first, last)        # observed: 0.118 -> 0.008
    assert after > before + 10.0, (before, after)  # observed: 22.1 -> 41.1 dB


def test_operator_path_training_fits_the_teacher_images():
    from sk_gs_amd.losses import image_loss
    from sk_gs_amd.optim import FusedAdam
    model, views, targets, bg, R = _scene()
    before = _psnr(model, views, targets, bg)
    opt = FusedAdam(model.param_groups(lr=2e-3), eps=1e-15)
    params = [p for p in model.parameters() if p.requires_grad]
    losses = []
    for it in range(120):
        v = it % V
        for p in params:
            p.grad = None
        loss = image_loss(model.render(views[v], time_id=v, background=bg)['images'], targets[v])
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu()
    first, last = float(losses[:8].mean()), float(losses[-8:].mean())
    after = _psnr(model, views, targets, bg)
    print(f'operator path: loss {first:.4f} -> {last:.4f}, PSNR {before:.2f} -> {after:.2f} dB')
    assert bool(torch.isfinite(losses).all())
    assert last < 0.3 * first, (first, last)       # observed: 0.118 -> 0.014
    assert after > before + 8.0, (before, after)  # observed: 22.1 -> 36.0 dB
