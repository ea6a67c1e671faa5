"""BASELINE's second metric and the reference's FPS protocol, measured beside the training rate (one rank).

ms/render: rasterizer forward + backward alone through the operator path with fixed upstream gradients; FPS: test.py:56-81,102-123
(warm-up, then N renders between two events): 20 warm-up + 200 timed iterations, HIP events on the launch stream.
"""
import torch

NW, NT = 20, 200


def ms_per_render(model, settings0, H, W, dev):
    """operator path render() + torch.autograd.grad of (images, opacity) w.r.t. its five inputs: eagerly (one HIP event pair per
    iteration, no host synchronisation) and as one captured graph replayed"""
    from sk_gs_amd.renderer.gaussian_render import render
    from sk_gs_amd.train_step import GraphedSteps
    with torch.no_grad():
        net = {k: v.detach() for k, v in model(0).items()}
    gcol, gop = torch.randn(3, H, W, device=dev), torch.randn(H, W, device=dev)
    ins = {k: v.clone().requires_grad_(True) for k, v in net.items()}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NT)]
    leaves = list(ins.values())

    def op_fwd_bwd(_=0):
        o = render(**ins, raster_settings=settings0)
        # (autograd.grad: the operator's gradients are RETURNED, not accumulated into leaf .grad tensors -- five AccumulateGrad
        # add kernels per iteration, 19 MB of them the SH gradient, are not the rasterizer)
        torch.autograd.grad([o['images'], o['opacity']], leaves, [gcol, gop], allow_unused=True)

    def timed(fn):
        for i in range(NW + NT):
            if i >= NW:
                ev[i - NW][0].record()
            fn(0)
            if i >= NW:
                ev[i - NW][1].record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)

    times = timed(op_fwd_bwd)
    out = dict(median=round(times[NT // 2], 4), p10=round(times[NT // 10], 4), p90=round(times[9 * NT // 10], 4),
               protocol=f'{NW} warm-up + {NT} timed iterations, one HIP event pair per iteration')
    # the same operator-path calls captured once and replayed (config.sync_num_rendered is off: nothing in them touches the host):
    # what the drop-in boundary costs without the Python / launch latency of the eager loop above
    try:
        g_op = GraphedSteps(op_fwd_bwd)
        g_op.capture(0)
        out['graph_replay_median'] = round(timed(g_op)[NT // 2], 4)
        del g_op
    except Exception as e:  # noqa  (a capture problem must not cost the headline line)
        out['graph_replay_median'] = None
        out['graph_replay_error'] = str(e)[:200]
    return out


def forward_fps(model, settings, frames, background, fused_forward=None):
    """forward-only render rate, the reference's FPS (deform network + skinning + rasterize + background, no_grad), through the
    operator path and -- `fused_forward(i)`: one graph replay of the fused step's forward for view i -- through the fused step"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for i in range(NW + NT):
            if i == NW:
                e0.record()
            model.render(settings[i % len(settings)], time_id=i % frames, background=background)
        e1.record()
    torch.cuda.synchronize()
    fps = dict(operator_path=round(NT * 1000.0 / e0.elapsed_time(e1), 1),
               protocol=f'test.py:102-123: {NW} warm-up + {NT} renders between two events, views cycled')
    if fused_forward is not None:
        for i in range(NW + NT):
            if i == NW:
                e0.record()
            fused_forward(i)
        e1.record()
        torch.cuda.synchronize()
        fps['fused_step_graph'] = round(NT * 1000.0 / e0.elapsed_time(e1), 1)
    return fps
