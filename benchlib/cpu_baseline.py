"""bench.py's `cpu_baseline` leg: the CPU oracle (oracle/: a restatement of the reference's kernels -- the reference has no CPU
path) timed on the host cores, on a bounded sample of the bench workload.

Two figures for the bench workload:
  value       deform + rasterize forward + backward (the oracle's kernels, OpenMP, all cores) -- the part of the step the
              reference implements in its extension;
  full_step   the same plus the rest of the training step on the host, so that the GPU line's `value` has a like-for-like
              baseline: 0.8 L1 + 0.2 (1 - SSIM) forward + backward (the torch restatement of networks/losses/ssim.py, CPU) and
              torch.optim.Adam(eps=1e-15) over every parameter tensor of the workload (8.45 M elements at config #1).
Beside them BASELINE config #0 (the reference's CPU-sized case) on all cores and on one thread.
"""
import os
import time

import torch

def _cpu_oracle():
    import tempfile
    from oracle import oracle as om
    lib = None
    try:
        out = om.build(tempfile.mkdtemp(prefix='skgs_oracle_native_'), native=True)
        lib = os.path.join(out, 'libskgs_oracle_native.so')
        if not os.path.exists(lib):
            lib = None
    except Exception:
        lib = None
    return om.Oracle('f32', lib_path=lib), lib


def _cpu_time(o, cfg, seconds_budget, max_iters=50):
    """(iterations, seconds) of deform + rasterize forward + backward of the oracle on the workload `cfg`"""
    from sk_gs_amd import scene, skeleton
    P, M, K, W, H = cfg['P'], cfg['M'], cfg['K'], cfg['W'], cfg['H']
    g = scene.make_gaussians(P, seed=0)
    cam = scene.make_camera(W, H, seed=0)
    rs = scene.raster_settings_from_camera(cam, colmap=True)
    n = lambda t: t.numpy()  # noqa: E731
    gen = torch.Generator().manual_seed(5)
    if M > 0:
        b = scene.make_bones(M, seed=0)
        table, _ = skeleton.build_ancestor_table(b['parents'], 0)
        sk_T = skeleton.kinematic(b['joints'], skeleton.axis_angle_to_quat(b['axis_angle']), None, table, 0)
        w = torch.softmax(torch.randn(P, K, generator=gen), -1)
    gcol = torch.randn(3, H, W, generator=gen).numpy()
    gop = torch.randn(H, W, generator=gen).numpy()

    def one_iter():
        if M > 0:
            _, idx = o.knn_bones(n(g['xyz']), n(b['joints']), K)
            d = o.lbs_deform_forward(n(g['xyz']), n(w), idx, n(sk_T), n(b['d_rot']), n(b['d_scale']), n(g['xyz']),
                                     n(g['log_scale']), n(g['rot']), n(g['opacity_logit']))
        else:
            act = scene.activate(g)
            d = dict(means=n(act['means3D']), scales=n(act['scales']), rotations=n(act['rotations']),
                     opacity=n(act['opacity']))
        fwd = o.rasterize_forward(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix), n(rs.projmatrix),
                                  n(rs.campos), d['means'], d['opacity'], n(g['sh']), d['scales'], d['rotations'])
        gr = o.rasterize_backward(fwd, H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, True, n(rs.viewmatrix), n(rs.projmatrix),
                                  n(rs.campos), d['means'], n(g['sh']), d['scales'], d['rotations'], gcol, gop)
        if M > 0:
            o.lbs_deform_backward(n(g['xyz']), n(w), idx, n(sk_T), n(b['d_rot']), n(b['d_scale']), n(g['log_scale']),
                                  n(g['rot']), n(g['opacity_logit']), gr['dL_dmeans3D'], gr['dL_dscales'],
                                  gr['dL_drotations'], gr['dL_dopacity'])

    one_iter()  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        one_iter()
        iters += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or iters >= max_iters:
            break
    return iters, el


def cpu_baseline(cfg, seconds_budget=20.0, single_thread_workload=False, configs=None):
    """time the CPU oracle (bounded samples); returns the cpu_baseline object.  The headline value is the bench workload on
    all host cores; beside it BASELINE config #0 (the reference's CPU-sized case) on all cores and on ONE thread, and --
    opt-in, it takes about a minute per iteration -- the bench workload on one thread."""
    o, lib = _cpu_oracle()
    cores = o.num_threads()
    iters, el = _cpu_time(o, cfg, 0.6 * seconds_budget)
    out = dict(value=round(iters / el, 4), unit='iters/s', cores=cores, kind='port',
               sample=f'{iters} iterations of deform+rasterize forward+backward (no loss/Adam) of the same workload, '
                      f'{el:.1f} s, oracle built {"-march=native" if lib else "portable"}, OpenMP')
    c0 = configs[0]
    i0, e0 = _cpu_time(o, c0, 0.15 * seconds_budget, max_iters=200)
    o.set_num_threads(1)
    i1, e1 = _cpu_time(o, c0, 0.25 * seconds_budget, max_iters=50)
    out['config0'] = dict(workload=f'{c0["name"]}: {c0["P"]} static Gaussians, {c0["W"]}x{c0["H"]}, rasterize forward+backward',
                          all_cores=dict(value=round(i0 / e0, 3), cores=cores, iterations=i0),
                          single_thread=dict(value=round(i1 / e1, 3), cores=1, iterations=i1), unit='iters/s')
    if single_thread_workload:
        i2, e2 = _cpu_time(o, cfg, 1.0, max_iters=1)
        out['single_thread'] = dict(value=round(i2 / e2, 5), cores=1, iterations=i2, unit='iters/s')
    o.set_num_threads(cores)
    try:
        out['full_step'] = _full_step(cfg, iters / el, min(6.0, 0.25 * seconds_budget))
    except Exception as e:  # noqa  (the extra leg must not cost the headline line)
        out['full_step'] = dict(error=f'{type(e).__name__}: {e}'[:200])
    return out


def _full_step(cfg, render_rate, seconds_budget):
    """loss forward + backward and Adam on the host, added to the oracle's deform + rasterize time per iteration"""
    from sk_gs_amd.losses import image_loss_torch
    P, M, W, H = cfg['P'], cfg['M'], cfg['W'], cfg['H']
    g = torch.Generator().manual_seed(11)
    pred = torch.rand(3, H, W, generator=g, requires_grad=True)
    gt = torch.rand(3, H, W, generator=g)
    # the parameter tensors of the workload: 59 floats per Gaussian + the LBS logits + the 8 x 256 deform network
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)] + ([(P, M)] if M > 0 else []) + [(1_100_000,)]
    params = [torch.nn.Parameter(torch.randn(*s_, generator=g) * 0.01) for s_ in shapes]
    opt = torch.optim.Adam(params, lr=1e-4, eps=1e-15)
    for p_ in params:
        p_.grad = torch.randn(p_.shape, generator=g) * 1e-3

    def one():
        loss = image_loss_torch(pred, gt)
        loss.backward()
        pred.grad = None
        opt.step()

    one()
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 20:
            break
    extra = (time.perf_counter() - t0) / n
    total = 1.0 / render_rate + extra
    return dict(value=round(1.0 / total, 4), unit='iters/s', loss_and_adam_s=round(extra, 4), iterations=n,
                threads=torch.get_num_threads(),
                sample=f'{n} iterations of L1 + SSIM forward + backward (torch CPU restatement) + torch.optim.Adam(eps=1e-15) over '
                       f'{sum(p_.numel() for p_ in params) / 1e6:.2f} M elements, added to the oracle\'s time per iteration')


