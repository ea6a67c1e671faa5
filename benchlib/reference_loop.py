"""`bench.py --reference-loop hooks | accelerated | fused`: one training iteration AS THE REFERENCE WRITES IT, on this package's hooks.

The GPU box has no /root/reference, so the loop is the reference's call sequence restated (benchlib/ref_sequence.py -- replayed against
the reference's own recorded run by tests/test_gpu_lietorch.py -- plus the lines of train.py / gaussian_splatting.py named below), written
only against what `sk_gs_amd.install_reference_hooks()` puts under an unmodified checkout:

    sk_deform_net(joints, t)                        SimpleDeformationNetwork: torch encoders + MLP_with_skips (sk_gs.py:134-164)
    kinematic / skeleton_warp_SE3                   lietorch stand-in (sk_gs.py:1069-1107, 193-206)
    calc_LBS_weight                                 pytorch3d.ops.knn_points stand-in + the reference's gather / softmax lines (:751-774)
    (sk_T[indices].act(points[:, None]) * w).sum    lietorch stand-in (:1147), d_rot / d_scale blends, the activations (:1192-1203)
    GaussianRasterizer(settings)(...)               diff_gaussian_rasterization stand-in, wxyz rotations (gaussian_render_origin.py:36-58)
    0.8 L1 + 0.2 (1 - SSIM)                         SSIM_Loss as torch convolutions (losses/ssim.py:20-62), image_loss.py:6-32
    loss.backward(); torch.optim.Adam(eps=1e-15).step(); zero_grad(set_to_none)            train.py:179-250, gaussian_splatting.py:443-453

`hooks`: exactly that -- every launch is one the reference's own Python issues (eager, no graph: the sequence holds blocking
host-to-device copies, `x.new_tensor([...])`).  `accelerated`: the same loop after `sk_gs_amd.accelerate_reference(fused_render=False)`
-- the five methods it patches (network forward, kinematic, calc_LBS_weight, SSIM_Loss.forward) and torch.optim.Adam.step run their fast
paths; everything else unchanged.  `fused` (round 6): the loop as `train.py:179-250` + `framework.execute_backward` write it --
`outputs = model.render(t=, info=, background=, time_id=)`, `losses = model.loss(...)` (the `rgb` / `ssim` lines, sk_gs.py:1524-1529,
through a `LossDict`), `sum(losses.values()).backward()`, `optimizer.step()`, `zero_grad(set_to_none=True)` -- on a stand-in model with
the reference's attribute names, with the three methods `accelerate_reference()` patches for the fused route bound to it:
`SkeletonGaussianSplatting.render`, `ImageLoss.forward`, `SSIM_Loss.forward` (sk_gs_amd/reference_fused.py).  `info` holds DEVICE
tensors, as after `tensor_to(data, device)` (train.py:180).
The number is what a user of the UNMODIFIED reference gets on one MI355X; the package's own trainer (`FusedTrainStep`, the default bench
line) is the same arithmetic as 12 launches in a graph.
"""
import os
import sys
import time
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def setup(args, configs):
    """everything up to the timed loop: the scene, the optimizer, the targets and ``step(i)`` of the requested mode (a namespace; the
    tests drive the pieces in-process: tests/test_gpu_reference_fused.py)"""
    from benchlib import ref_sequence as rs
    from sk_gs_amd import _C, lietorch as L, pytorch3d_ops as p3d, reference_accel as ra, scene
    from sk_gs_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sk_gs_amd.losses import ssim_loss
    from sk_gs_amd.skeleton import build_ancestor_table
    sys.modules.setdefault('lietorch', L)
    fused = args.reference_loop == 'fused'
    accel = args.reference_loop == 'accelerated' or fused
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.autograd.set_multithreading_enabled(False)
    _C.load_library()
    _C.config.sync_num_rendered = True    # the reference's forward blocks on num_rendered (gaussian_rasterizer_forward.cu:209); the hook's default
    cfg = configs[args.config]
    P, M, K, W, H = cfg['P'], cfg['M'], cfg['K'], cfg['W'], cfg['H']
    assert M > 0, '--reference-loop needs a skinned config (1..4)'
    sp = args.stage == 'sp'       # stage sp: 512 superpoints, search over xyz + 8 hyper dimensions, sp_deform_net (sk_gs.py:830-856)
    if sp:
        M, K = args.superpoints, args.knn
    frames = args.views
    gs, bones = scene.make_gaussians(P, seed=0, sh_degree=3, scale_mult=args.scale_mult), scene.make_bones(M, seed=0)
    if args.loop_scene == 'headline' and not sp:   # Gaussians in Z-order, as the default line keeps them (densify.sort_spatially)
        from sk_gs_amd.densify import morton_order
        order = morton_order(gs['xyz'])
        gs = {k_: v_[order].contiguous() for k_, v_ in gs.items()}
    table, _ = build_ancestor_table(bones['parents'].long(), 0)
    g = torch.Generator().manual_seed(0)
    par = lambda t: torch.nn.Parameter(t.clone().to(dev))  # noqa: E731
    p = dict(_xyz=par(gs['xyz']), _features_dc=par(gs['sh'][:, :1]), _features_rest=par(gs['sh'][:, 1:]), _scaling=par(gs['log_scale']),
             _rotation=par(gs['rot']), _opacity=par(gs['opacity_logit']), sp_W=par(torch.randn(P, M, generator=g)),
             joints=par(bones['joints']), global_tr=par(torch.tensor([[0., 0, 0, 0, 0, 0, 1]]).repeat(frames, 1)))
    torch.manual_seed(1)
    if sp:
        from sk_gs_amd.densify import morton_order
        from sk_gs_amd.superpoint import SpDeformNet
        pick = torch.randperm(P, generator=g)[:M]
        pick = pick[morton_order(gs['xyz'][pick])]
        for k_ in ('sp_W', 'joints', 'global_tr'):
            p.pop(k_)
        p.update(sp_points=par(gs['xyz'][pick]), hyper_feature=par(torch.full((P, 8), -1e-2) + 0.02 * torch.randn(P, 8, generator=g)),
                 sp_hyper_feature=par(torch.full((M, 8), 1e-2) + 0.02 * torch.randn(M, 8, generator=g)),
                 _sp_radius=par(torch.full((M,), -1.35)), _sp_weight=par(torch.zeros(M)))
        # --lbs-method: which parameters of the weighting exist decides the weighting (sk_gs.py:464-475,757-770)
        if args.lbs_method == 'W':
            p.pop('_sp_radius'), p.pop('_sp_weight')
            p['sp_W'] = par(torch.randn(P, M, generator=g))
        elif args.lbs_method == 'kernel':
            p.pop('_sp_weight')
        elif args.lbs_method == 'dist':
            p.pop('_sp_radius'), p.pop('_sp_weight')
        # DeformNetwork's structure and parameter names (sk_gs.py:209-315; pinned by tests/golden/sp_deformnet.npz) + the attributes
        # the accelerator probes on the reference's class
        net = SpDeformNet(sep_rot=bool(args.sep_rot))
        net.pos_enc_p, net.pos_enc_t, net.max_d_scale = rs.RefFreqEncoder(3, 10), rs.RefFreqEncoder(1, 6), -1.0
        net = net.to(dev)
        with torch.no_grad():
            net.gaussian_warp.weight.normal_(0, 2e-3), net.gaussian_rotation.weight.normal_(0, 2e-3), net.gaussian_scaling.weight.normal_(0, 2e-5)
            if args.sep_rot:
                net.local_rotation.weight.normal_(0, 2e-3)
        ra._originals.setdefault('sp_net', lambda self, x, t, **kw: dict(self.reference_forward(x, t)))
    else:
        net = rs.RefSimpleDeformationNetwork().to(dev)
        with torch.no_grad():
            if args.loop_scene == 'headline':   # the default bench line's model: start of the skeleton stage (sk_gs_amd/model.py)
                for h in net.dynamic_net.last:
                    h.weight.mul_(0.01), h.bias.zero_()
            else:                               # (round 5's loop scene: larger joint rotations)
                for h, s in zip(net.dynamic_net.last, (0.2, 1e-2, 1e-3)):
                    h.weight.normal_(0, s / 16.)
    if accel and 'adam' not in ra._originals:   # (what accelerate_reference(adam=True) does)
        ra._originals['adam'] = torch.optim.Adam.step
        torch.optim.Adam.step = ra.adam_step
    lr = args.lr
    groups = [
        {'params': [p['_xyz']], 'lr': lr * 0.16, 'name': 'xyz'}, {'params': [p['_features_dc']], 'lr': lr * 2.5, 'name': 'f_dc'},
        {'params': [p['_features_rest']], 'lr': lr * 2.5 / 20, 'name': 'f_rest'}, {'params': [p['_opacity']], 'lr': lr * 50., 'name': 'opacity'},
        {'params': [p['_scaling']], 'lr': lr * 5.0, 'name': 'scaling'}, {'params': [p['_rotation']], 'lr': lr, 'name': 'rotation'}]
    if sp:   # get_params of stage sp (sk_gs.py:583-602)
        groups += [{'params': [p['hyper_feature']], 'lr': lr * 2.5, 'name': 'hyper'}, {'params': list(net.parameters()), 'lr': lr * 0.16, 'name': 'sp_deform'},
                   {'params': [p['sp_points']], 'lr': lr * 0.16, 'name': 'sp_points'}] + \
                  [{'params': [p[k_]], 'lr': lr * 0.16, 'name': n_} for k_, n_ in (('_sp_radius', 'sp_radius'), ('_sp_weight', 'sp_weight'), ('sp_W', 'sp_W')) if k_ in p] + \
                  [{'params': [p['sp_hyper_feature']], 'lr': lr * 2.5, 'name': 'sp_hyper'}]
    else:
        groups += [{'params': [p['sp_W']], 'lr': lr, 'name': 'sp_W'}, {'params': [p['global_tr']], 'lr': lr, 'name': 'skinning'},
                   {'params': list(net.parameters()), 'lr': lr, 'name': 'deform_net'}, {'params': [p['joints']], 'lr': lr * 0.1, 'name': 'joints'}]
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    times = torch.linspace(0., 1., frames, device=dev).view(frames, 1)
    bg = torch.ones(3, device=dev)
    settings = []
    for v in range(args.views):
        r = scene.raster_settings_from_camera(scene.make_camera(W, H, seed=v), sh_degree=3, colmap=True, device=dev)
        settings.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=r.tanfovx, tanfovy=r.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=r.viewmatrix,
            projmatrix=r.projmatrix, sh_degree=3, campos=r.campos, prefiltered=False, debug=False))
    # a stand-in `self` for the two patched methods of SkeletonGaussianSplatting (what tests/test_gpu_lietorch.py uses)
    me = types.SimpleNamespace(training=True, test_time_interpolate=False, sk_feature=None, _R_dim=4, joint_parents=table.to(dev).int(),
                               joint_root=torch.tensor(0), sk_cache=torch.zeros(frames, M, 11, device=dev), num_knn=K,
                               _sp_radius=p.get('_sp_radius'), _sp_weight=p.get('_sp_weight'), sp_W=p.get('sp_W'), sk_is_init=torch.tensor(True),
                               sk_deform_net=lambda x, t: ra.simple_deform_forward(net, x, t))
    if accel:
        ra._originals.setdefault('sk_net', rs.RefSimpleDeformationNetwork.forward)
    ssim_self = types.SimpleNamespace(window_size=11, reduction='mean')

    def deform_sp(v):
        points = p['_xyz'].detach()
        if not accel or args.warp_method != 'LBS' or args.sep_rot:   # (the per-method fast paths below restate the LBS lines only)
            out = net.reference_forward(p['sp_points'].detach(), times[v])              # the module's own torch forward
            a = dict(p, net_d_xyz=out['d_xyz'], net_d_rotation=out['d_rotation'], net_d_scaling=out['d_scaling'])
            if args.sep_rot:
                a['net_g_rotation'] = out['g_rotation']
            return rs.sp_stage(L, p3d.knn_points, a, K, args.warp_method, bool(args.sep_rot))
        if '_sp_radius' in p:                                                                  # the reference's properties (:548-553)
            me.kernel_radius = torch.exp(p['_sp_radius'])
        if '_sp_weight' in p:
            me.kernel_weight = torch.sigmoid(p['_sp_weight'])
        w, idx = ra.calc_LBS_weight(me, points, p['sp_points'], p['hyper_feature'], p['sp_hyper_feature'])
        out = ra.deform_network_forward(net, p['sp_points'].detach(), times[v])
        bias = points.new_tensor([0, 0, 0, 1.])                                              # (sk_gs.py:835: the reference's own line)
        d_rot = F.normalize(out['d_rotation'] + bias, dim=-1)
        spT = L.SE3.InitFromVec(torch.cat([out['d_xyz'], d_rot], dim=-1))
        d_points = (spT[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
        res = rs._activate(p, d_points, (d_rot[idx] * w[..., None]).sum(dim=1), (out['d_scaling'][idx] * w[..., None]).sum(dim=1))
        res['_knn_w'], res['_spT'] = w, spT.vec()
        return res

    def deform(v):
        if sp:
            return deform_sp(v)
        points = p['_xyz'].detach()
        a = dict(p, time_id=torch.tensor(v), parents_table=table.to(dev), root=torch.tensor(0))
        if not accel:
            a['net_sk_r'], a['net_d_rot'], a['net_d_scale'] = net(p['joints'], times[v])
            return rs.sk_stage(L, p3d.knn_points, a, K)
        sk_T, sk_d_rot, sk_d_scale = ra.kinematic(me, p['joints'], times[v], p['global_tr'][v].view(-1), v, None)
        w, idx = ra.calc_LBS_weight(me, points, p['joints'])
        d_xyz = (sk_T[idx].act(points[:, None]) * w[..., None]).sum(dim=1) - points
        return rs._activate(a, d_xyz, (sk_d_rot[idx] * w[..., None]).sum(dim=1), (sk_d_scale[idx] * w[..., None]).sum(dim=1))

    def render(v, res):
        means2D = torch.zeros_like(res['points'], requires_grad=True)                     # gaussian_splatting.py: screenspace_points
        shs = torch.cat([p['_features_dc'], p['_features_rest']], dim=1)
        img, _ = GaussianRasterizer(settings[v])(means3D=res['points'], means2D=means2D, opacities=res['opacity'], shs=shs,
                                                 scales=res['scales'], rotations=(ra.QuatXYZW.wrap(res['rotations']) if accel else res['rotations'])[..., (3, 0, 1, 2)])   # the adapter's swizzle (gaussian_render_origin.py:41-42)
        return img

    def loss_of(img, gt):
        l1 = (img - gt).abs().mean()
        if accel:
            ss = ra.ssim_loss_forward(ssim_self, img, gt)
        else:
            ss = ssim_loss(img, gt)
        return 0.8 * l1 + 0.2 * ss

    ra._originals.setdefault('ssim', lambda self, a, b: ssim_loss(a, b))
    with torch.no_grad():
        gen = torch.Generator().manual_seed(77)
        targets = [(render(v, deform(v)) + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1) for v in range(args.views)]

    def step(i):
        v = i % args.views
        opt.zero_grad(set_to_none=True)
        loss = loss_of(render(v, deform(v)), targets[v])
        loss.backward()
        opt.step()
        return loss

    # the two regularisers of the shipped sp configuration on the [P,K] LBS weights (--sp-regularisers), as the reference writes them:
    # loss_weight_sparsity / loss_weight_smooth (sk_gs.py:1339-1359) over a 20-neighbour table of the Gaussians (gs_knn_num, :345)
    gs_knn_index = None
    if sp and args.sp_regularisers:
        assert args.reference_loop != 'hooks', '--sp-regularisers: accelerated | fused'
        with torch.no_grad():
            pts = p['_xyz'].detach()
            gs_knn_index = p3d.knn_points(pts[None], pts[None], None, None, K=21)[1][0].contiguous()      # (pykdtree in the reference, :1348-1353)

    reg_self = types.SimpleNamespace(gs_knn_index=gs_knn_index, update_gs_knn=lambda: None)
    ra._originals.setdefault('w_sparse', lambda self, w, eps=1e-7: -(w * torch.log(w + eps) + (1 - w) * torch.log(1 - w + eps)).mean())   # sk_gs.py:1339-1340
    ra._originals.setdefault('w_smooth', lambda self, w: (w[:, None] - w[self.gs_knn_index]).abs().mean())                                # :1357-1359

    def weight_regularisers(knn_w, reference=False):     # sk_gs.py:1572-1574 with the weights of exps/default.yaml:85-86
        if reference or args.reg_torch:                  # as the reference writes them (torch)
            sparse, smooth = ra._originals['w_sparse'](reg_self, knn_w), ra._originals['w_smooth'](reg_self, knn_w[0])
        else:                                            # as accelerate_reference() patches the two methods (one launch each)
            sparse, smooth = ra.loss_weight_sparsity(reg_self, knn_w), ra.loss_weight_smooth(reg_self, knn_w[0])
        return 0.1 * sparse + 0.1 * smooth

    if gs_knn_index is not None and not fused:
        base_step = step

        def step(i):  # noqa: F811
            v = i % args.views
            opt.zero_grad(set_to_none=True)
            res = deform(v)
            loss = loss_of(render(v, res), targets[v]) + weight_regularisers(res['_knn_w'][None])
            loss.backward()
            opt.step()
            return loss

    rf = None
    if fused:
        from sk_gs_amd import reference_fused as rf
        if sp:
            model = _RefSuperpointModel(p, net, K)
            model.warp_method, model.sep_rot = args.warp_method, bool(args.sep_rot)
        else:
            model = _RefSkeletonModel(p, net, table.to(dev).int(), frames, M, K, dev)
        # what a call outside the fused route's conditions reaches: the reference's own render, i.e. the `accelerated` sequence above
        ra._originals['render'] = lambda self, *a, t=None, info=None, time_id=None, **kw: {
            'images': render(int(time_id), deform(int(time_id))).permute(1, 2, 0)[None], 'stage': args.stage}
        image_crit, ssim_crit = _RefImageLoss(), types.SimpleNamespace(window_size=11, reduction='mean')
        ra._originals['image_loss'] = _RefImageLoss.reference_forward
        cams = [scene.make_camera(W, H, seed=v) for v in range(args.views)]
        infos = [{k: (c[k].to(dev) if torch.is_tensor(c[k]) else c[k]) for k in ('Tw2v', 'Tv2c', 'FoV', 'campos', 'size')} for c in cams]
        time_ids = [torch.tensor([v], device=dev) for v in range(args.views)]               # int64, as the loader's `time_id`
        targets_hwc = [tg.permute(1, 2, 0)[None].contiguous() for tg in targets]             # [1,H,W,3]: targets['images']
        weights = {'image': 0.8, 'ssim': 0.2}                                                # exps/default.yaml:86-88

        def loss_funcs(name, *inputs):                                                       # LossDict.forward (losses/build.py:55-64)
            func = {'image': lambda a, b: rf.image_loss_forward(image_crit, a, b), 'ssim': lambda a, b: ra.ssim_loss_forward(ssim_crit, a, b)}[name]
            return func(*inputs) * weights[name]

        def model_loss(outputs, tgt):                                                        # sk_gs.py:1524-1529 (stage sk: nothing else)
            image, gt_img = outputs['images'], tgt[..., :3]
            Hh, Ww, Cc = image.shape[-3:]
            image, gt_img = image.view(1, Hh, Ww, Cc), gt_img.view(1, Hh, Ww, Cc)
            losses = {'rgb': loss_funcs('image', image, gt_img), 'ssim': loss_funcs('ssim', image, gt_img)}
            if gs_knn_index is not None:                                                      # :1572-1574
                losses['sparse+smooth'] = weight_regularisers(outputs['_knn_w'])
            return losses

        def step(i):  # noqa: F811
            v = i % args.views
            outputs = rf.render(model, t=times[v], info=infos[v], background=bg, time_id=time_ids[v], stage=args.stage)      # train.py:190
            losses = model_loss(outputs, targets_hwc[v])                                      # :194
            loss = sum(losses.values())                                                      # framework.py:268
            loss.backward()                                                                  # :286
            opt.step()                                                                       # :304
            opt.zero_grad(set_to_none=True)                                                  # :305
            return loss

    return types.SimpleNamespace(step=step, p=p, net=net, opt=opt, rf=rf, ra=ra, L=L, p3d=p3d, _C=_C, cfg=cfg, P=P, M=M, K=K, W=W, H=H, sp=sp,
                                 accel=accel, fused=fused, targets=targets, deform=deform, render=render, loss_of=loss_of, times=times, bg=bg,
                                 model=locals().get('model'), infos=locals().get('infos'), time_ids=locals().get('time_ids'),
                                 targets_hwc=locals().get('targets_hwc'), model_loss=locals().get('model_loss'), weight_regularisers=weight_regularisers)


def run(args, configs):
    s = setup(args, configs)
    step, rf, ra, L, p3d, _C, cfg, P, M, K, W, H, sp, accel, fused, model = (
        s.step, s.rf, s.ra, s.L, s.p3d, s._C, s.cfg, s.P, s.M, s.K, s.W, s.H, s.sp, s.accel, s.fused, s.model)
    # untimed prime steps, as the headline run does and for its reason (--prime-steps: a fresh process and a fresh scene reach a training
    # run's steady state only after some tens of steps -- here also the tile lists of the synthetic scene's first steps, which the route
    # re-measures at calls 32 and 128); reported as `prime_steps`
    prime = max(int(args.prime_steps), 0)
    # every mode: CPython's oldest-generation collection walks all of torch (~100 ms, every ~200 eager iterations); the set-up's objects
    # are frozen out of it, as the fused route does for itself (sk_gs_amd/reference_fused.py::_freeze_collector_once)
    import gc
    gc.collect()
    gc.freeze()
    for i in range(prime + max(args.warmup, 5)):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(prime + args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = 1e3 * elapsed / args.steps
    return {
        'metric': f"train iters/sec of the REFERENCE's own call sequence on the hooks ({args.reference_loop}), {P // 1000}k Gaussians @{W}x{H}",
        'value': round(args.steps / elapsed, 3), 'unit': 'iters/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': max(args.warmup, 5), 'prime_steps': prime,
        'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{cfg["name"]}' + (' in stage sp' if sp else '') + f': {P} Gaussians, {M} ' + ('superpoints (3+8-d search, weighted_kernel)' if sp else 'bones') + f', K={K}, SH degree 3, {W}x{H}, {args.views} synthetic views',
                   'step': ("the reference's call sequence restated (benchlib/ref_sequence.py + train.py:179-250) on install_reference_hooks() alone: "
                            "torch network, lietorch / pytorch3d stand-ins, diff_gaussian_rasterization stand-in, torch SSIM, torch.optim.Adam; eager"
                            if not accel else
                            "the same loop after accelerate_reference(): network forward, kinematic, calc_LBS_weight and SSIM_Loss.forward on "
                            "their fast paths, torch.optim.Adam.step as one launch (sk_gs_amd.reference_accel); rasterizer stand-in unchanged; eager"
                            if not fused else
                            "train.py:179-250 / framework.execute_backward restated -- outputs = model.render(t, info, background, time_id); "
                            "losses = model.loss(...) [rgb, ssim through a LossDict]; sum(losses).backward(); optimizer.step(); zero_grad(set_to_none) "
                            "-- with SkeletonGaussianSplatting.render, ImageLoss.forward and SSIM_Loss.forward as accelerate_reference() patches them "
                            "(sk_gs_amd.reference_fused: the package's fused launches on the model's own Parameters, info on the device, "
                            "no host read-back), torch.optim.Adam.step as one launch; eager"),
                   'loop_scene': args.loop_scene, 'loss_last': float(last.detach()), 'sync_num_rendered': bool(_C.config.sync_num_rendered), 'accelerators': dict(ra.calls) if accel else None,
                   'lie_fused_calls': dict(L.fused_calls), 'knn_hip_calls': dict(p3d.hip_calls),
                   'fused_route': None if rf is None else dict(calls=dict(rf.calls), why_not=dict(rf.why_not),
                                                                status=_route_status(rf, model))},
    }


def _route_status(rf, model):
    r = rf.route_of_model(model, 'sp' if getattr(model, '_stage', 'sk') == 'sp' else 'sk')
    if r is None:
        return None
    st = r.step.status()
    return dict(tile_bucket=r._bucket, overflow_events=st['overflow_events'], mlp_failed=st.get('mlp_failed', 0))


class _RefImageLoss:
    """ImageLoss(method='l1') (networks/losses/image_loss.py:6-32): the attributes the patch probes + the reference's own forward"""
    method, masked = 'l1', False

    def reference_forward(self, pred_image, gt_image, mask=None):
        return F.l1_loss(pred_image[..., :3], gt_image[..., :3])


class _RefSuperpointModel:
    """the attributes of ``SkeletonGaussianSplatting`` the fused route reads in stage sp (networks/sk_gs.py:342-540)"""
    training, use_official_gaussians_render, convert_SHs_python, compute_cov3D, _stage = True, True, False, False, 'sp'
    warp_method, sep_rot, max_sh_degree, hyper_dim = 'LBS', False, 3, 8

    def __init__(self, p, net, K):
        for k_ in ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_points', 'hyper_feature', 'sp_hyper_feature'):
            setattr(self, k_, p[k_])
        self.sp_W, self._sp_radius, self._sp_weight = p.get('sp_W'), p.get('_sp_radius'), p.get('_sp_weight')
        self.LBS_method = 'W' if self.sp_W is not None else ('weighted_kernel' if self._sp_weight is not None else
                                                              'kernel' if self._sp_radius is not None else 'dist')
        self.sp_deform_net, self.num_knn = net, K
        self._active_sh_degree = torch.tensor(3, dtype=torch.int, device=p['_xyz'].device)
        self.sp_weights = self.sp_knn = None

    def get_now_stage(self, stage=None):
        return 'sp' if stage is None else stage


class _RefSkeletonModel:
    """the attributes of ``SkeletonGaussianSplatting`` (networks/sk_gs.py:342-540, gaussian_splatting.py:101-183) the fused route reads,
    under the reference's names, around the SAME Parameter objects the optimizer holds (tests/test_host_cpu.py runs the route's condition
    and re-homing code on the reference's real class in the build container)"""
    training, use_official_gaussians_render, convert_SHs_python, compute_cov3D = True, True, False, False
    LBS_method, sk_feature, _R_dim, max_sh_degree, test_time_interpolate = 'W', None, 4, 3, False

    def __init__(self, p, net, table, frames, M, K, dev):
        for k_ in ('_xyz', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'sp_W', 'joints', 'global_tr'):
            setattr(self, k_, p[k_])
        self.sk_deform_net, self.joint_parents, self.joint_root, self.num_knn = net, table, torch.tensor(0), K
        self.sk_cache = torch.zeros(frames, M, 11, device=dev)
        self.sk_is_init = torch.tensor(True, device=dev)
        self._active_sh_degree = torch.tensor(3, dtype=torch.int, device=dev)

    def get_now_stage(self, stage=None):
        return 'sk' if stage is None else stage
