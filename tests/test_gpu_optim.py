"""Fused multi-tensor Adam kernel vs torch.optim.Adam (same hyper-parameters as the reference: eps 1e-15).
Tolerance: 1e-6 max-norm relative on parameters and both moments after 6 steps."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch():
    from sk_gs_amd.optim import FusedAdam
    from sk_gs_amd.view_parallel import FlatGradBuffer
    gen = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (37,), (5, 7), (4099,)]  # incl. unaligned offsets
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 1e-3, 1e-3, 1e-3]
    a = [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam([{'params': [p], 'lr': lr} for p, lr in zip(a, lrs)], eps=1e-15, betas=(0.9, 0.999))
    fb = FlatGradBuffer(b)  # gradients as views of one flat buffer (some of them not 16-byte aligned)
    opt = FusedAdam([{'params': [p], 'lr': lr} for p, lr in zip(b, lrs)], eps=1e-15, betas=(0.9, 0.999))
    for step in range(6):
        for p, q in zip(a, b):
            g = torch.randn(p.shape, generator=gen).cuda() * (0.0 if step == 3 else 1.0)  # one all-zero gradient step
            p.grad = g.clone()
            q.grad.copy_(g)
        ref.step()
        opt.step()
    assert float(opt.step_count.item()) == 6.0
    for p, q in zip(a, b):
        assert rel_err(q, p) <= 1e-6
        assert rel_err(opt.state[q]['exp_avg'], ref.state[p]['exp_avg']) <= 1e-6
        assert rel_err(opt.state[q]['exp_avg_sq'], ref.state[p]['exp_avg_sq']) <= 1e-6
    opt.set_lr(0, 0.5)
    before = b[0].detach().clone()
    opt.step()
    assert float((b[0] - before).abs().max()) > 1e-3


def test_fused_adam_clears_the_requested_gradient_span_after_the_update():
    """zero_after_step: the launch that bumps the step counter also clears a gradient span (FusedViewStep's per-frame
    table gradients), after the update has consumed it"""
    from sk_gs_amd.optim import FusedAdam
    flat = torch.ones(10 + 37, device='cuda')
    a, b = torch.nn.Parameter(torch.zeros(10, device='cuda')), torch.nn.Parameter(torch.zeros(37, device='cuda'))
    a.grad, b.grad = flat[:10], flat[10:]
    opt = FusedAdam([{'params': [a, b], 'lr': 0.1}], zero_after_step=flat[10:])
    opt.step()
    torch.cuda.synchronize()
    assert float(a.grad.min()) == 1.0 and float(b.grad.abs().max()) == 0.0
    assert float(b.data.max()) < 0.0 and float(opt.step_count.item()) == 1.0  # b was updated with the gradient first


def test_fused_adam_state_dict_round_trip_and_torch_layout():
    """FusedAdam.state_dict / load_state_dict in torch.optim.Adam's layout: a reloaded optimizer continues exactly where
    the saved one stood (moments AND step counter: the bias correction goes on), and a state saved by torch.optim.Adam over
    the same groups loads too"""
    import copy
    from sk_gs_amd.optim import FusedAdam
    torch.manual_seed(0)

    def make():
        a = torch.nn.Parameter(torch.randn(1000, 3, device='cuda'))
        b = torch.nn.Parameter(torch.randn(77, device='cuda'))
        return a, b, [dict(params=[a], lr=1e-2, name='a'), dict(params=[b], lr=3e-3, name='b')]

    a, b, groups = make()
    opt = FusedAdam(groups, eps=1e-15)
    ref = torch.optim.Adam([dict(params=[a.detach().clone().requires_grad_(True)], lr=1e-2),
                            dict(params=[b.detach().clone().requires_grad_(True)], lr=3e-3)], eps=1e-15)
    grads = [(torch.randn_like(a), torch.randn_like(b)) for _ in range(5)]
    for ga, gb in grads[:3]:
        a.grad, b.grad = ga.clone(), gb.clone()
        opt.step()
        for p, g in zip([q for grp in ref.param_groups for q in grp['params']], (ga, gb)):
            p.grad = g.clone()
        ref.step()
    sd = opt.state_dict()
    tsd = ref.state_dict()
    assert sorted(sd['state'].keys()) == sorted(tsd['state'].keys()) == [0, 1]
    for i in (0, 1):
        assert float(sd['state'][i]['step']) == float(tsd['state'][i]['step']) == 3
        assert rel_err(sd['state'][i]['exp_avg'], tsd['state'][i]['exp_avg']) <= 1e-6     # (torch's lerp vs an fma: an ulp)
        assert rel_err(sd['state'][i]['exp_avg_sq'], tsd['state'][i]['exp_avg_sq']) <= 1e-6
    assert [g['params'] for g in sd['param_groups']] == [[0], [1]] and sd['param_groups'][0]['name'] == 'a'
    # fresh optimizer over copies of the current parameters + the saved state = the original, step for step
    a2 = torch.nn.Parameter(a.detach().clone())
    b2 = torch.nn.Parameter(b.detach().clone())
    opt2 = FusedAdam([dict(params=[a2], lr=5.0, name='a'), dict(params=[b2], lr=5.0, name='b')], eps=1e-15)
    opt2.load_state_dict(copy.deepcopy(sd))
    assert float(opt2.step_count) == 3 and opt2.param_groups[0]['lr'] == 1e-2
    for ga, gb in grads[3:]:
        a.grad, b.grad, a2.grad, b2.grad = ga.clone(), gb.clone(), ga.clone(), gb.clone()
        opt.step(), opt2.step()
    assert torch.equal(a, a2) and torch.equal(b, b2)
    # a torch.optim.Adam state loads into the fused optimizer
    a3 = torch.nn.Parameter(a.detach().clone())
    b3 = torch.nn.Parameter(b.detach().clone())
    opt3 = FusedAdam([dict(params=[a3], lr=1.0, name='a'), dict(params=[b3], lr=1.0, name='b')], eps=1e-15)
    opt3.load_state_dict(tsd)
    assert float(opt3.step_count) == 3
    assert torch.allclose(opt3.state[a3]['exp_avg'], tsd['state'][0]['exp_avg'].cuda())


def test_a_step_taken_in_pieces_equals_the_whole_step():
    """``step(groups, advance=False)`` pieces + ``advance_step()`` (skgs_adam_step_range): every piece uses the bias
    correction of the same step, the counter moves once -- bit-identical to one launch over all groups"""
    from sk_gs_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(1)
    names = ['xyz', 'f_dc', 'f_rest', 'opacity', 'net']
    shapes = [(3000, 3), (3000, 1, 3), (3000, 15, 3), (3000, 1), (257, 33)]

    def make():
        g = torch.Generator().manual_seed(2)
        ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
        return ps, FusedAdam([{'params': [p], 'lr': 1e-2 * (i + 1), 'name': n} for i, (p, n) in enumerate(zip(ps, names))])
    (a, whole), (b, pieces) = make(), make()
    span = torch.ones(64, device='cuda')
    pieces.zero_after_step = span
    for _ in range(4):
        for p, q in zip(a, b):
            g = torch.randn(p.shape, generator=gen).cuda()
            p.grad.copy_(g), q.grad.copy_(g)
        whole.step()
        pieces.step(['f_dc', 'f_rest'], advance=False)
        pieces.step(['xyz', 'opacity'], advance=False)  # not neighbours in the table: two launches
        assert float(span.sum()) == 64.0 or _ > 0          # a piece does not clear the span ...
        pieces.step(['net'], advance=False)
        pieces.advance_step()
    assert float(pieces.step_count.item()) == 4.0 and float(span.abs().sum()) == 0.0  # ... the closing call does
    for p, q in zip(a, b):
        assert torch.equal(p, q)
        assert torch.equal(whole.state[p]['exp_avg_sq'], pieces.state[q]['exp_avg_sq'])
    with pytest.raises(AssertionError):
        pieces.step(['no_such_group'])


def test_adam_rows_as_the_side_job_of_the_network_backward_launch():
    """``skgs_deform_mlp_backward_adam``: the update of a run of parameter groups applied by the extra workgroups of the deform
    network's backward launch is bit-identical to ``step(groups, advance=False)``, the network's own gradients are unchanged,
    and the counter does not move until the closing piece"""
    from sk_gs_amd.deform_net import DeformMLP, FusedDeformMLP
    from sk_gs_amd.optim import FusedAdam
    torch.manual_seed(0)
    mlp = DeformMLP().cuda()
    B = 20
    joints, t, g = torch.rand(B, 3, device='cuda') - 0.5, torch.tensor([0.3], device='cuda'), torch.randn(B, 11, device='cuda')
    net = mlp.dynamic_net
    params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
    names = ['xyz', 'f_dc', 'f_rest', 'opacity', 'sp_W', 'net']
    shapes = [(30011, 3), (30011, 1, 3), (30011, 15, 3), (30011, 1), (30011, 20), (129, 7)]  # odd sizes: ragged last chunks

    def make():
        gen = torch.Generator().manual_seed(2)
        ps = [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in shapes]
        opt = FusedAdam([{'params': [p], 'lr': 1e-2 * (i + 1), 'name': n} for i, (p, n) in enumerate(zip(ps, names))])
        return ps, opt
    (a, ref), (b, side) = make(), make()
    rows = names[:5]
    run = FusedDeformMLP(mlp, B)
    gen = torch.Generator().manual_seed(3)
    for it in range(3):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=gen).cuda()
            p.grad.copy_(gr), q.grad.copy_(gr)
        run.forward(joints, t)
        g1 = [torch.zeros_like(p) for p in params]
        run.backward(joints, t, g, g1)                                  # the launch alone
        ref.step(rows, advance=False)
        ref.step(['net'])
        g2 = [torch.zeros_like(p) for p in params]
        run.backward(joints, t, g, g2, side_adam=side.side_range(rows))  # the launch with the rows' update on board
        assert float(side.step_count.item()) == float(it)
        side.step(['net'])
        for x, y in zip(g1, g2):
            assert torch.equal(x, y)
    assert run.status()['failed'] == 0 and float(side.step_count.item()) == 3.0
    for p, q in zip(a, b):
        assert torch.equal(p, q)
        assert torch.equal(ref.state[p]['exp_avg'], side.state[q]['exp_avg'])
        assert torch.equal(ref.state[p]['exp_avg_sq'], side.state[q]['exp_avg_sq'])


@pytest.mark.parametrize('in_backward', [0.6, 0.0])
def test_rows_split_between_the_backward_launch_and_the_next_forward_launch(in_backward):
    """the pre-forward schedule of ``FusedTrainStep``: 60 % of the rows' chunks beside the skeleton backward, the closing piece
    (counter moves), the other 40 % beside the NEXT skeleton-forward launch with ``after_advance`` -- bit-identical to one
    ``step()``, and the forward launch's own outputs (heads, bone transforms) are those of the launch alone.  0.0: the
    view-parallel schedule (``reduce_between``), every row beside the forward"""
    from sk_gs_amd.deform_net import BoneChainDesc, FusedDeformMLP
    from sk_gs_amd.model import SkinnedGaussians
    from sk_gs_amd.optim import FusedAdam
    torch.manual_seed(0)
    M = 20
    model = SkinnedGaussians(500, M, 4, sh_degree=0, num_frames=3, seed=5, deform_net=True, learn_joints=True).cuda()
    mlp, topo = model.sk_deform_net, model.topology()
    joints, t = model.joints.detach().contiguous(), torch.tensor([0.37], device='cuda')
    gT = model.global_tr.detach()[1].contiguous()
    f32 = dict(dtype=torch.float32, device='cuda')
    net = mlp.dynamic_net
    params = [p for l in net.net for p in (l.weight, l.bias)] + [net.last_weight, net.last_bias]
    gh = [torch.randn((M, 4), **f32), torch.randn((M, 4), **f32), torch.randn((M, 3), **f32)]
    grads, gx = [torch.zeros_like(p) for p in params], torch.zeros(M, net.in_channels, **f32)
    g_bone_T, g_j, g_g = torch.randn(M, 7, **f32), torch.zeros(M, 3, **f32), torch.zeros(7, **f32)

    def outputs():
        heads = [torch.empty((M, 4), **f32), torch.empty((M, 4), **f32), torch.empty((M, 3), **f32)]
        bone_T, chain_A = torch.zeros(M, 7, **f32), torch.zeros(M, 7, **f32)
        b = BoneChainDesc()
        b.M, b.root, b.num_levels = M, topo['root'], topo['num_levels']
        b.parents, b.level_nodes, b.level_start = (topo[k].data_ptr() for k in ('parents', 'level_nodes', 'level_start'))
        b.joints, b.global_T, b.bone_T, b.chain_A = joints.data_ptr(), gT.data_ptr(), bone_T.data_ptr(), chain_A.data_ptr()
        b.sk_r_raw, b.g_bone_T, b.g_joints, b.g_global_T = heads[0].data_ptr(), g_bone_T.data_ptr(), g_j.data_ptr(), g_g.data_ptr()
        return heads, bone_T, chain_A, b
    names = ['xyz', 'f_dc', 'f_rest', 'opacity', 'sp_W', 'net']
    shapes = [(30011, 3), (30011, 1, 3), (30011, 15, 3), (30011, 1), (30011, 20), (129, 7)]

    def make():
        gen = torch.Generator().manual_seed(2)
        ps = [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in shapes]
        opt = FusedAdam([{'params': [p], 'lr': 1e-2 * (i + 1), 'name': n} for i, (p, n) in enumerate(zip(ps, names))])
        return ps, opt
    (a, ref), (b_, split) = make(), make()
    rows = names[:5]
    run = FusedDeformMLP(mlp, M)
    alone, riding = outputs(), outputs()
    run.forward(joints, t, head_out=alone[0], bones=alone[3])
    gen = torch.Generator().manual_seed(3)
    for it in range(3):
        for p, q in zip(a, b_):
            gr = torch.randn(p.shape, generator=gen).cuda()
            p.grad.copy_(gr), q.grad.copy_(gr)
        ref.step()
        head = split.side_range(rows, (0.0, in_backward)) if in_backward else None
        tail = split.side_range(rows, (in_backward, 1.0), after_advance=True)
        assert tail.chunk_end == split._chunk_ranges(rows)[0][1]
        assert tail.chunk_begin == (head.chunk_end if head is not None else split._chunk_ranges(rows)[0][0])
        run.backward(joints, t, gh, grads, gx, bones=riding[3], side_adam=head)
        split.step_tail(['net'])
        assert float(split.step_count.item()) == it + 1.0
        run.forward(joints, t, head_out=riding[0], bones=riding[3], side_adam=tail)
        for x, y in zip(alone[0] + list(alone[1:3]), riding[0] + list(riding[1:3])):
            assert torch.equal(x, y)
        for p, q in zip(a, b_):
            assert torch.equal(p, q), it
            assert torch.equal(ref.state[p]['exp_avg'], split.state[q]['exp_avg'])
            assert torch.equal(ref.state[p]['exp_avg_sq'], split.state[q]['exp_avg_sq'])
    assert run.status()['failed'] == 0


def test_closing_piece_with_the_encoder_backward_in_the_joints_workgroup():
    """the tail of a fused step: rows as the side job of the network's backward launch, then ``skgs_adam_step_tail`` over
    tables + network + joints with the frequency-encoding backward run by the workgroup that updates the joints -- against
    backward, ``skgs_freq_encode_backward`` and ONE ``FusedAdam.step`` as separate launches: bit-identical"""
    import copy
    from sk_gs_amd.deform_net import DeformMLP, DeformMLPRunner, FusedDeformMLP
    from sk_gs_amd.optim import FusedAdam
    torch.manual_seed(0)
    B = 20
    mlps = [DeformMLP().cuda()]
    with torch.no_grad():
        mlps[0].dynamic_net.last_weight.normal_(0, 0.1)
    mlps.append(copy.deepcopy(mlps[0]))
    joints0 = torch.rand(B, 3, device='cuda') - 0.5
    t, g = torch.tensor([0.3], device='cuda'), torch.randn(B, 11, device='cuda')
    gen = torch.Generator().manual_seed(2)
    rows0 = [torch.randn(20011, 3, generator=gen).cuda(), torch.randn(20011, 15, 3, generator=gen).cuda()]
    tab0 = torch.randn(6, 7, generator=gen).cuda()
    sides = []
    for mlp in mlps:
        net = mlp.dynamic_net
        layers = [(l.weight, l.bias) for l in net.net] + [(net.last_weight, net.last_bias)]
        rows = [torch.nn.Parameter(r.clone()) for r in rows0]
        joints, tab = torch.nn.Parameter(joints0.clone()), torch.nn.Parameter(tab0.clone())
        opt = FusedAdam([{'params': [rows[0]], 'lr': 1e-3, 'name': 'xyz'}, {'params': [rows[1]], 'lr': 2e-3, 'name': 'f_rest'},
                         {'params': [tab], 'lr': 1e-3, 'name': 'skinning'},
                         {'params': list(mlp.parameters()), 'lr': 1e-3, 'name': 'deform_net'},
                         {'params': [joints], 'lr': 1e-4, 'name': 'joints'}])
        opt.zero_after_step = tab.grad.view(-1)
        sides.append(dict(mlp=mlp, layers=layers, rows=rows, joints=joints, tab=tab, opt=opt, run=FusedDeformMLP(mlp, B),
                          gx=torch.zeros(B, net.in_channels, device='cuda')))
    ref, fus = sides
    for it in range(3):
        gr = [torch.randn(r.shape, generator=gen).cuda() for r in rows0]
        gt, gj = torch.randn(tab0.shape, generator=gen).cuda(), torch.randn(joints0.shape, generator=gen).cuda()
        for sd in sides:
            for p, x in zip(sd['rows'], gr):
                p.grad.copy_(x)
            sd['tab'].grad.copy_(gt), sd['joints'].grad.copy_(gj)
            sd['run'].forward(sd['joints'].detach(), t)
            sd['grads'] = [x.grad for l in sd['layers'] for x in l]
        # separate launches
        ref['run'].backward(ref['joints'].detach(), t, g, ref['grads'], ref['gx'])
        DeformMLPRunner(ref['mlp']).input_grad(ref['gx'], ref['run'].x0, ref['joints'].grad, accumulate=True)
        ref['opt'].step()
        # rows inside the backward launch + the closing launch
        o, m = fus['opt'], fus['mlp']
        fus['run'].backward(fus['joints'].detach(), t, g, fus['grads'], fus['gx'], side_adam=o.side_range(['xyz', 'f_rest']))
        o.step_tail(['skinning', 'deform_net', 'joints'], freq_param=fus['joints'],
                    freq_job=(B, m.p_in, m.p_degree, fus['gx'], fus['run'].x0, fus['run'].x0.shape[1], fus['joints'].grad, True))
        assert float(o.step_count.item()) == it + 1 and float(fus['tab'].grad.abs().sum()) == 0.0
    assert fus['run'].status()['failed'] == 0
    for a, b in zip(ref['opt'].params, fus['opt'].params):
        assert torch.equal(a, b)
        assert torch.equal(fus['opt'].state[b]['exp_avg'], ref['opt'].state[a]['exp_avg'])
        assert torch.equal(fus['opt'].state[b]['exp_avg_sq'], ref['opt'].state[a]['exp_avg_sq'])


def test_captured_steps_with_gradients_handed_over_by_autograd():
    """``p.grad = None`` in front of a CAPTURED backward (zero_grad(set_to_none=True) semantics: autograd stores each gradient
    as it is, no zero fill and no "+=" launch): the gradients live in the graph's pool, at other addresses in every graph.
    ``FusedAdam.step`` gives each capture a descriptor table of its own; two graphs replayed alternately, a learning-rate
    change between replays and an eager step in between all follow torch.optim.Adam on a replica."""
    from sk_gs_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(3)
    shapes, lrs = [(1000, 3), (257,), (33, 15, 3)], [1e-2, 3e-3, 1e-3]
    a = [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    coef = [[torch.randn(*s, generator=gen).cuda() for s in shapes] for _ in range(2)]  # two "views"
    ref = torch.optim.Adam([{'params': [p], 'lr': lr} for p, lr in zip(a, lrs)], eps=1e-15)
    opt = FusedAdam([{'params': [p], 'lr': lr, 'name': f'g{i}'} for i, (p, lr) in enumerate(zip(b, lrs))], eps=1e-15)

    def loss(params, v):
        return sum(((p * p) * c).sum() + (p * c).sum() for p, c in zip(params, coef[v])) + (params[1][:5] ** 3).sum()

    def one(v):  # the captured body
        for p in b:
            p.grad = None
        loss(b, v).backward()
        opt.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graphs = []
    with torch.cuda.stream(side):
        pool = None
        for v in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=side):
                one(v)
            pool = g.pool()
            graphs.append(g)
    torch.cuda.current_stream().wait_stream(side)
    # capturing runs nothing; the two graphs share a pool, so their gradients may land at the SAME addresses -- then they
    # share one descriptor table (tables are keyed by the addresses they describe)
    assert float(opt.step_count.item()) == 0.0 and len(opt._capture_tables) in (1, 2)

    def ref_step(v):
        ref.zero_grad(set_to_none=True)
        loss(a, v).backward()
        ref.step()

    order = [0, 1, 1, 0, 1, 0]
    for i, v in enumerate(order):
        if i == 3:
            opt.set_lr('g0', 5e-2)
            ref.param_groups[0]['lr'] = 5e-2
        if i == 4:  # an eager step between replays (fresh gradients outside a capture: the bound table is refreshed)
            one(0)
            ref_step(0)
        graphs[v].replay()
        ref_step(v)
    torch.cuda.synchronize()
    assert float(opt.step_count.item()) == len(order) + 1
    for p, q in zip(a, b):
        assert rel_err(q, p) <= 2e-6
        assert rel_err(opt.state[q]['exp_avg'], ref.state[p]['exp_avg']) <= 2e-6


def test_captured_tables_are_recycled_across_recaptures_and_optimizer_surgery():
    """the descriptor tables of captured steps used to come from a 64-slot arena that was never
    released -- a training loop that re-captures after every densification (~145 events in the reference schedule) died
    after 64.  Now: pieces of one step share a slot, surgery that re-creates tensors (change_optimizer) recycles every slot
    and re-sizes the arenas when the parameter COUNT changes, `release_captured_tables` frees them on request, and a
    learning-rate change reaches captured steps in stream order.  200 recaptures, each followed by replays, follow
    torch.optim.Adam on a replica."""
    from sk_gs_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(11)
    a = [torch.nn.Parameter(torch.randn(300, 3, generator=gen).cuda()), torch.nn.Parameter(torch.randn(77, generator=gen).cuda())]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam([{'params': [a[0]], 'lr': 1e-2}, {'params': [a[1]], 'lr': 3e-3}], eps=1e-15)
    opt = FusedAdam([{'params': [b[0]], 'lr': 1e-2, 'name': 'rows'}, {'params': [b[1]], 'lr': 3e-3, 'name': 'vec'}], eps=1e-15)
    opt.MAX_CAPTURED_TABLES = 4  # a small arena makes exhaustion immediate if nothing is recycled
    opt.release_captured_tables()
    assert opt._capture_slots == 4

    def loss(params):
        return ((params[0] ** 2).sum(1) * 0.5).sum() + (params[1] ** 3).sum()

    side = torch.cuda.Stream()

    def capture(pieces):
        g = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for p in opt.params:
                    p.grad = None
                loss(opt.params).backward()
                if pieces:  # a step taken in pieces: both pieces see the same gradients -> ONE table
                    opt.step(groups=['rows'], advance=False)
                    opt.step(groups=['vec'], advance=True)
                else:
                    opt.step()
        torch.cuda.current_stream().wait_stream(side)
        return g

    def ref_step():
        ref.zero_grad(set_to_none=True)
        loss([g['params'][0] for g in ref.param_groups]).backward()
        ref.step()

    steps = 0
    for it in range(12):
        g = capture(pieces=bool(it % 2))
        assert len(opt._capture_tables) == 1, 'pieces of one captured step share a table'
        for _ in range(2):
            g.replay()
            ref_step()
            steps += 2 - 1
        if it == 5:
            opt.set_lr('vec', 1e-2)  # reaches the captured table in stream order
            ref.param_groups[1]['lr'] = 1e-2
            g.replay()
            ref_step()
            steps += 1
        # "densification": the row tensor is re-created together with its moments -> every captured table is stale
        keep = torch.ones(opt.params[0].shape[0], dtype=torch.bool, device='cuda')
        keep[it] = False
        del g
        opt.change_optimizer(keep, 'rows', op='prune')
        assert len(opt._capture_tables) == 0, 'surgery that moves addresses recycles the captured tables'
        rg = ref.param_groups[0]
        old = rg['params'][0]
        st = ref.state.pop(old)
        new = torch.nn.Parameter(old.data[keep].clone())
        rg['params'][0] = new
        ref.state[new] = {'step': st['step'], 'exp_avg': st['exp_avg'][keep].clone(), 'exp_avg_sq': st['exp_avg_sq'][keep].clone()}
    torch.cuda.synchronize()
    assert float(opt.step_count.item()) == steps
    for q, grp in zip(opt.params, ref.param_groups):
        p = grp['params'][0]
        assert rel_err(q, p) <= 5e-6
        assert rel_err(opt.state[q]['exp_avg'], ref.state[p]['exp_avg']) <= 5e-6
    # exhaustion is an actionable error, and a release makes room again
    graphs = [capture(False) for _ in range(1)]
    with pytest.raises(RuntimeError, match='release_captured_tables'):
        for _ in range(8):  # (each capture's gradients land at new pool addresses while the earlier graphs are alive)
            graphs.append(capture(False))
    torch.cuda.synchronize()
    del graphs
    opt.release_captured_tables(slots=16)
    assert opt._capture_slots == 16 and not opt._capture_tables
    g = capture(False)
    g.replay()
    torch.cuda.synchronize()


@pytest.mark.parametrize('P,M,K', [(3000, 512, 5), (257, 100, 7), (64, 1024, 16)])
def test_sparse_logit_table_adam_is_bit_identical_to_the_dense_step(P, M, K):
    """LBS_method 'W' (sk_gs.py:469-471, exps/default.yaml:35): skgs_adam_logit_rows -- the Adam update of the dense [P, M] logit
    table restricted to the 32-column tiles a row has ever received a gradient in, the gradient formed on the fly from the
    step's neighbours -- against the dense path (skgs_lbs_weights_backward writes the [P, M] gradient, FusedAdam steps over all
    of it): parameters and both moments BIT-identical over steps whose neighbour sets drift; the tile mask rebuilt from the
    moments continues identically."""
    import ctypes as C
    from sk_gs_amd import _C
    from sk_gs_amd.optim import FusedAdam
    lib = _C.load_library()
    dev = 'cuda'
    g = torch.Generator().manual_seed(P + M)
    w0 = torch.randn(P, M, generator=g)
    other0 = torch.randn(1000, generator=g)

    def make():
        spw, other = torch.nn.Parameter(w0.clone().to(dev)), torch.nn.Parameter(other0.clone().to(dev))
        opt = FusedAdam([{'params': [other], 'lr': 1e-2, 'name': 'other'}, {'params': [spw], 'lr': 3e-3, 'name': 'sp_W'}], eps=1e-15)
        other.grad.normal_(generator=torch.Generator(device=dev).manual_seed(1))
        return spw, other, opt

    spw_a, other_a, opt_a = make()   # dense
    spw_b, other_b, opt_b = make()   # sparse
    mask = torch.zeros(P, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    score = torch.rand(P, M, generator=g)
    for step in range(7):
        score = score + 0.35 * torch.rand(P, M, generator=g)   # the neighbour sets drift
        idx = score.topk(K, dim=1).indices.to(dev).contiguous()
        w = torch.softmax(torch.randn(P, K, generator=g), -1).to(dev)
        gw = torch.randn(P, K, generator=g).to(dev)
        if step == 3:
            gw[::3] = 0.0   # rows whose gradient is exactly zero: their moments only decay
        _C._check(lib.skgs_lbs_weights_backward(C.c_int32(P), C.c_int32(M), C.c_int32(K), p(w), p(idx), p(gw), p(spw_a.grad), _C._stream()))
        opt_a.step()
        opt_b.step(['other'], advance=False)
        _C._check(lib.skgs_adam_logit_rows(C.c_int32(P), C.c_int32(M), C.c_int32(K), p(w), p(idx), p(gw),
                                           C.c_void_p(opt_b.table_entry(spw_b)), p(mask), C.c_double(0.9), C.c_double(0.999),
                                           C.c_double(1e-15), C.c_void_p(opt_b.step_state.data_ptr()), C.c_int32(0), _C._stream()))
        opt_b.advance_step()
        if step == 4:  # the mask is a cache of "a moment of this tile is non-zero": rebuilt, the run continues identically
            rebuilt = torch.zeros_like(mask)
            st = opt_b.state[spw_b]
            _C._check(lib.skgs_adam_logit_mask_rebuild(C.c_int32(P), C.c_int32(M), p(st['exp_avg']), p(st['exp_avg_sq']), p(rebuilt), _C._stream()))
            assert bool(((rebuilt & ~mask) == 0).all())   # nothing live outside the accumulated mask
            mask.copy_(rebuilt)
        for name, a, b in (('param', spw_a, spw_b), ('exp_avg', opt_a.state[spw_a]['exp_avg'], opt_b.state[spw_b]['exp_avg']),
                           ('exp_avg_sq', opt_a.state[spw_a]['exp_avg_sq'], opt_b.state[spw_b]['exp_avg_sq']), ('other', other_a, other_b)):
            same = torch.equal(a.detach(), b.detach())
            assert same, f'step {step} {name}: max diff {float((a.detach() - b.detach()).abs().max()):.3e}, ' \
                         f'{int((a.detach() != b.detach()).sum())} elements differ'
        assert float(opt_a.step_count) == float(opt_b.step_count) == step + 1
    tiles = (M + 31) // 32
    live = sum(int(((mask >> t) & 1).sum()) for t in range(tiles)) / (P * tiles)
    assert live < 1.0 or K * 7 >= tiles  # (the point of the kernel: most tiles are never visited)


def test_state_listeners_hear_every_change_of_the_moments_from_outside_a_step():
    from sk_gs_amd.optim import FusedAdam
    a, b = torch.nn.Parameter(torch.randn(50, 3, device='cuda')), torch.nn.Parameter(torch.randn(7, device='cuda'))
    opt = FusedAdam([{'params': [a], 'lr': 1e-2, 'name': 'xyz'}, {'params': [b], 'lr': 1e-2, 'name': 'b'}], eps=1e-15)
    heard, once = [], []
    opt.add_state_listener(lambda: heard.append(1) or True)
    opt.add_state_listener(lambda: once.append(1) or False)       # returns False: dropped after its first call
    a.grad.normal_(), b.grad.normal_()
    opt.step()
    assert not heard
    opt.load_state_dict(opt.state_dict())
    assert len(heard) == 1 and len(once) == 1
    keep = torch.ones(50, dtype=torch.bool, device='cuda')
    keep[::5] = False
    opt.change_optimizer(keep, 'xyz', op='prune')
    assert len(heard) == 2 and len(once) == 1
    rows = torch.arange(40, device='cuda', dtype=torch.int32)
    opt.gather_rows(['xyz'], rows, 40)
    assert len(heard) == 3


@pytest.mark.parametrize('n_big', [0, 3_000_000])
def test_device_lr_schedules_follow_the_reference_step_for_step_inside_multi_step_graphs(n_big):
    """update_learning_rate runs before every train step in the reference (train.py:140-141); here the launch that
    advances the step counter evaluates get_expon_lr_func on the device.  Four optimizer steps per hipGraph replay over 40 steps: the
    rate of EVERY step equals float32 of the reference's value (tests/golden/lr_schedule_dense.npz, written by the reference's own
    get_expon_lr_func) bit for bit, and it is the rate the update applies; restored step counts (a resumed run) and the schedules'
    stage offset land on the right values too.  n_big: a large tensor beside them -- the step then has more than 256 chunks and the
    counter is advanced by the separate bump launch instead of the update launch's last workgroup."""
    import os
    import numpy as np
    from sk_gs_amd.optim import FusedAdam, position_lr
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lr_schedule_dense.npz'))
    steps = z['steps'].tolist()
    want = {name: {s: np.float32(v) for s, v in zip(steps, z[name])} for name in ('xyz', 'deform', 'eased')}
    args = {name: z[name + '_args'] for name in want}
    dev = 'cuda'
    xyz, net, other = (torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (3000, 700, 50))
    groups = [{'params': [xyz], 'lr': 123.0, 'name': 'xyz'}, {'params': [other], 'lr': 1e-2, 'name': 'other'},
              {'params': [net], 'lr': 456.0, 'name': 'sk_deform'}]
    if n_big:
        groups.insert(1, {'params': [torch.nn.Parameter(torch.zeros(n_big, device=dev))], 'lr': 1e-3, 'name': 'big'})
    opt = FusedAdam(groups, eps=1e-15)

    def sched(name, group, offset=0):
        a = args[name]
        opt.set_lr_schedule(group, lr_init=a[0], lr_final=a[1], lr_delay_steps=int(a[2]), lr_delay_mult=a[3], max_steps=int(a[4]), step_offset=offset)
    sched('xyz', 'xyz')
    sched('eased', 'sk_deform')
    for p in (xyz, net, other):
        p.grad.fill_(0.5)                                  # constant gradient: m_hat / sqrt(v_hat) = 1, every step moves by its rate
    assert np.float32(opt.scheduled_lr('xyz')) == want['xyz'][1] and np.float32(opt.scheduled_lr('sk_deform')) == want['eased'][1]
    assert opt.scheduled_lr('other') == 1e-2
    # four steps per replay; after each step the state's rates (of the NEXT step) and the parameters are logged by the graph itself
    log_lr, log_p = torch.zeros(4, 8, device=dev), torch.zeros(4, 3, device=dev)
    g = torch.cuda.CUDAGraph()
    opt.rebind()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            pass
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for i in range(4):
            opt.step()
            log_lr[i].copy_(opt.step_state[8:16])
            log_p[i].copy_(torch.stack([xyz.detach()[0], net.detach()[0], other.detach()[0]]))
    # (the capture itself does not execute: the counter is still 0)
    assert float(opt.step_count) == 0
    prev = np.zeros(3, np.float64)
    for r in range(10):
        g.replay()
        torch.cuda.synchronize()
        lr_next, p_now = log_lr.cpu().numpy(), log_p.cpu().numpy().astype(np.float64)
        for i in range(4):
            step = 4 * r + i + 1                              # the 1-based training step just taken
            assert np.float32(lr_next[i, 0]) == want['xyz'][step + 1], (step, lr_next[i, 0], want['xyz'][step + 1])
            assert np.float32(lr_next[i, 1]) == want['eased'][step + 1], (step, lr_next[i, 1], want['eased'][step + 1])
            moved = prev - p_now[i]
            for j, rate in enumerate((want['xyz'][step], want['eased'][step], np.float32(1e-2))):
                assert abs(moved[j] - float(rate)) <= 3e-6 * float(rate) + 1e-9, (step, j, moved[j], rate)
            prev = p_now[i]
    assert float(opt.step_count) == 40
    # a restored step count: the rates are re-derived for it (float32 of the reference's double, every sampled step up to 45 000)
    bad = 0
    for name, group in (('xyz', 'xyz'), ('eased', 'sk_deform')):
        for st in steps[65::7]:
            opt._set_step_count(float(st - 1))               # st - 1 steps taken: the next one is training step st
            got = np.float32(opt.scheduled_lr(group))
            bad += int(got != want[name][st])
            assert abs(float(got) - float(want[name][st])) <= 1.2e-7 * float(want[name][st]), (name, st)
    assert bad == 0, f'{bad} sampled steps differ from float32(reference) in the last bit'
    # the stage offset of sk_gs.py:621-626, and the host twin used with set_lr
    opt._set_step_count(0.0)
    sched('deform', 'sk_deform', offset=10_000)
    opt._set_step_count(10_088.0)
    assert np.float32(opt.scheduled_lr('sk_deform')) == want['deform'][89]
    a = args['deform']
    assert np.float32(position_lr(89, a[0], a[1], int(a[4]), int(a[2]), a[3])) == want['deform'][89]
    opt.clear_lr_schedules()
    assert opt.scheduled_lr('xyz') == 123.0


def test_device_lr_schedule_reaches_every_kind_of_update_piece():
    """a scheduled group's rate is the same in the one-launch step and in a step taken in pieces -- against torch.optim.Adam fed the
    reference schedule by hand (pieces that run AFTER the counter advanced, FusedTrainStep's pre-forward, use the closed step's rates:
    tests/test_gpu_bench_step.py trains with the schedule on, pre-forward on and off)"""
    from sk_gs_amd.optim import FusedAdam, position_lr
    dev = 'cuda'
    g = torch.Generator().manual_seed(3)
    a0, b0 = torch.randn(5000, generator=g), torch.randn(9000, generator=g)
    grads = [(torch.randn(5000, generator=g), torch.randn(9000, generator=g)) for _ in range(6)]
    kw = dict(lr_init=2e-2, lr_final=1e-4, max_steps=5, lr_delay_steps=3, lr_delay_mult=0.2)

    def run(mode):
        a, b = torch.nn.Parameter(a0.clone().to(dev)), torch.nn.Parameter(b0.clone().to(dev))
        if mode == 'torch':
            opt = torch.optim.Adam([{'params': [a], 'lr': 1.0}, {'params': [b], 'lr': 3e-3}], eps=1e-15)
        else:
            opt = FusedAdam([{'params': [a], 'lr': 1.0, 'name': 'xyz'}, {'params': [b], 'lr': 3e-3, 'name': 'rest'}], eps=1e-15)
            opt.set_lr_schedule('xyz', **kw)
        for i, (ga, gb) in enumerate(grads):
            if mode == 'torch':
                opt.param_groups[0]['lr'] = position_lr(i + 1, kw['lr_init'], kw['lr_final'], kw['max_steps'], kw['lr_delay_steps'], kw['lr_delay_mult'])
                a.grad, b.grad = ga.to(dev), gb.to(dev)
                opt.step()
            else:
                a.grad.copy_(ga), b.grad.copy_(gb)
                if mode == 'one':
                    opt.step()
                else:
                    opt.step(['rest'], advance=False)
                    opt.step(['xyz'], advance=True)
        torch.cuda.synchronize()
        return a.detach().cpu(), b.detach().cpu()
    ref = run('torch')
    for mode in ('one', 'pieces'):
        got = run(mode)
        assert rel_err(got[0], ref[0]) <= 5e-6 and rel_err(got[1], ref[1]) <= 5e-6, (mode, rel_err(got[0], ref[0]))
    assert torch.equal(run('one')[0], run('pieces')[0])


def test_accelerated_torch_adam_step_equals_torch():
    """accelerate_reference(adam=True): ``torch.optim.Adam.step`` of the optimizer the reference builds (ten groups, eps 1e-15, a rate per
    group that changes every iteration) as ONE launch over the optimizer's own state tensors -- parameters and both moments against an
    untouched torch.optim.Adam over six steps, with the reference's kind of state surgery in between (a parameter and its moments
    replaced by longer tensors, gaussian_splatting.py:cat_tensors_to_optimizer); steps outside the conditions (first step, a missing
    gradient) are torch's own"""
    from sk_gs_amd import reference_accel as ra
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(0)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 15, 3), (5000, 1), (5000, 3), (5000, 4), (5000, 20), (8, 7), (256, 84), (256,), (20, 3)]

    def build():
        params = [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(i)).to(dev)) for i, s in enumerate(shapes)]
        groups = [{'params': [p], 'lr': 1e-3 * (1 + i), 'name': f'g{i}'} for i, p in enumerate(params[:8])]
        groups.append({'params': params[8:10], 'lr': 1e-4, 'name': 'net'})
        groups.append({'params': params[10:], 'lr': 1e-5, 'name': 'joints'})
        return params, torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    pa, oa = build()
    pb, ob = build()
    original = torch.optim.Adam.step
    before = dict(ra.calls)
    try:
        ra._originals['adam'] = original
        for it in range(6):
            grads = [torch.randn(p.shape, generator=g).to(dev) * (1e-3 if it % 2 else 1.0) for p in pa]
            for p, q, gr in zip(pa, pb, grads):
                p.grad, q.grad = gr.clone(), gr.clone()
            if it == 4:
                pa[3].grad = pb[3].grad = None       # a parameter without a gradient this step: skipped, its counter stays
            for o in (oa, ob):
                o.param_groups[0]['lr'] = 1e-3 * 0.9 ** it      # update_learning_rate (train.py:140-141)
            ra.adam_step(oa)
            original(ob)
            if it == 2:   # densification: parameter 0 and its state grow (the optimizer's state dict is edited in place, as the reference does)
                for params, o in ((pa, oa), (pb, ob)):
                    old = params[0]
                    st = o.state.pop(old)
                    new = torch.nn.Parameter(torch.cat([old.detach(), torch.ones(100, 3, device=dev)]))
                    st['exp_avg'] = torch.cat([st['exp_avg'], torch.zeros(100, 3, device=dev)])
                    st['exp_avg_sq'] = torch.cat([st['exp_avg_sq'], torch.zeros(100, 3, device=dev)])
                    o.param_groups[0]['params'] = [new]
                    o.state[new] = st
                    params[0] = new
        torch.cuda.synchronize()
    finally:
        ra._originals.pop('adam', None)
    # (step 0 creates the state; step 4 skips the parameter without a gradient as torch does; at step 5 that parameter's counter is one
    # behind the others: a launch of its own with its own bias corrections)
    assert ra.calls['adam_fused'] - before['adam_fused'] == 5 and ra.calls['adam_reference'] - before['adam_reference'] == 1
    for i, (p, q) in enumerate(zip(pa, pb)):
        assert p.shape == q.shape
        assert rel_err(p, q) <= 2e-6, (i, rel_err(p, q))
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa['step']) == float(sb['step'])
        assert rel_err(sa['exp_avg'], sb['exp_avg']) <= 2e-6 and rel_err(sa['exp_avg_sq'], sb['exp_avg_sq']) <= 2e-6


def test_rescheduling_a_group_frees_its_old_schedule_slot():
    """``set_lr_schedule`` only appended -- the reference's piecewise schedule (offset 0, then sp_fix[0], then sk_init[0], for
    `xyz` and for the deform groups: six distinct entries over a run, sk_gs.py:619-626) hit 'at most 4' at the second stage boundary.
    The table is rebuilt from the entries still in use: the three stage boundaries of a run fit, and the rates stay the reference's."""
    import math
    from sk_gs_amd.optim import FusedAdam
    dev = torch.device('cuda')
    ps = [torch.nn.Parameter(torch.ones(64, device=dev)) for _ in range(3)]
    opt = FusedAdam([{'params': [ps[0]], 'lr': 1e-3, 'name': 'xyz'}, {'params': [ps[1]], 'lr': 1e-3, 'name': 'sp_deform'},
                     {'params': [ps[2]], 'lr': 1e-3, 'name': 'sk_deform'}], eps=1e-15)
    for offset in (40_000, 13_000, 0):        # the three offsets of a run (sk_init, sp_fix, static: exps/default.yaml:12-19), the last one checked below
        opt.set_lr_schedule('xyz', 1.6e-4 * 5, 1.6e-6 * 5, 30_000, 0, 0.01, step_offset=offset)
        opt.set_lr_schedule(['sp_deform', 'sk_deform'], 8e-4, 8e-6, 40_000, 0, 0.01, step_offset=offset)
        assert len(opt._schedules) == 2 and sorted(opt._sched_of_group.values()) == [0, 1, 1]
    # a fifth / sixth distinct entry IN USE at the same time is still refused
    opt.set_lr_schedule('sp_deform', 1e-3, 1e-5, 100, 0, 1.0)
    assert len(opt._schedules) == 3
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    torch.cuda.synchronize()
    want = math.exp(math.log(8e-4) * (1 - 2 / 40_000) + math.log(8e-6) * (2 / 40_000))     # delay_steps = 0: the plain log-lerp at step 2
    assert abs(opt.scheduled_lr('sk_deform') - want) <= 1e-6 * want
