"""bench.py prints exactly one JSON line on stdout with the fields the driver reads (short run, config #1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '8', '--warmup', '2', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 8 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-4 and r['achieved'] > 0
    ws = r['whole_step']  # the whole step against the roofline, per-kernel fractions, the cluster record
    assert ws['alg_bytes_render'] > 1e8 and abs(ws['ms'] - d['ms_per_step']) < 1e-3 and 0 < ws['frac_render_only'] < ws['frac'] < 1
    assert 'traffic' in r and 'traffic_source' in r
    assert d['config']['cluster']['world'] == 1 and d['config']['cluster']['devices'][0]['name']
    assert abs(d['value'] - 1000.0 / d['ms_per_step']) / d['value'] < 0.01
    b = d['ms_per_step_blocks']  # the spread of the timed region: 8 steps = 2 replays of the 4-step graph -> 2 blocks
    assert d['steps_per_graph'] == 4 and d['prime_steps'] == 100 and '4 consecutive steps per replay' in d['config']['launch']
    assert b['blocks'] == 2 and b['min'] <= b['p10'] <= b['median'] <= b['p90'] <= b['max']
    assert abs(b['median'] - d['ms_per_step']) / d['ms_per_step'] < 0.25
    # every launch of the step is timed and priced (algorithmic bytes next to the 8 TB/s roofline); together they are the step
    k = d['kernels']
    # (the skinning and its backward are jobs of the rasterizer's per-Gaussian launches)
    assert 'deform_forward' in k['preprocess_forward']['includes'] and 'deform_backward' in k['preprocess_backward']['includes']
    for name in ('preprocess_forward', 'scatter', 'tile_sort', 'render_forward', 'render_backward', 'preprocess_backward',
                 'image_loss_forward', 'image_loss_backward', 'skeleton_forward', 'skeleton_backward', 'adam'):
        assert k[name]['us'] > 0 and k[name]['alg_MB'] > 0 and 0 < k[name]['GBps'] < 8000.0, (name, k[name])
        assert abs(k[name]['frac'] - k[name]['GBps'] / 8000.0) < 1e-3
    total = sum(v['us'] * v['launches_per_step'] for v in k.values())
    assert 0.8 * d['ms_per_step'] * 1e3 < total < 1.6 * d['ms_per_step'] * 1e3  # (eager, event-bracketed: a little over)


PLAIN = ['--exchange', 'allreduce']


@pytest.mark.parametrize('port,ranks,extra', [(29577, 2, PLAIN), (29578, 2, ['--pipeline']), (29579, 2, ['--compact-logits']),
                                              (29580, 2, ['--autograd']), (29581, 4, PLAIN), (29582, 2, ['--sh-factors']),
                                              (29583, 2, ['--sh-factors', '--overlap-gather']), (29584, 2, PLAIN + ['--bone-tables']),
                                              (29587, 2, PLAIN + ['--graph-per-view']), (29588, 2, PLAIN + ['--pre-forward', 'off']),
                                              (29589, 8, PLAIN)])
def test_ranks_share_the_gpu_and_stay_identical(port, ranks, extra):
    """The driver's multi-GPU launch line with 2, 4 or 8 ranks (the driver's largest run: no rank-count assumption may surface on
    the first real 8-GPU lease) on this one GPU (gloo moves the gradients: RCCL refuses two ranks
    per device): the real view-parallel schedule -- graph(fwd+bwd) | all-reduce | graph(Adam) -- must keep the replicas
    bit-identical and print one line from rank 0 with whole-job throughput.  `--exchange allreduce`: ONE plain all-reduce of
    the flat gradient buffer; the byte-saving / overlapping exchanges are flags (the default, `auto`, ranks them: next test)."""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '6',
           '--warmup', '2', '--no-cpu-baseline'] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == ranks and d['steps'] == 6 and d['scaling'] == 'weak'
    assert d['config']['replicas_identical'] is True
    par = d['config']['parallelism']
    if extra == PLAIN:
        assert d['config']['exchange'] == 'allreduce'
        assert 'flat-buffer grad all-reduce' in par and 'compact' not in par and 'factors' not in par, par
        assert 'closing launch' in d['config']['adam'] and 'skeleton-forward' in d['config']['adam'], d['config']['adam']
    if extra == PLAIN + ['--pre-forward', 'off']:
        assert d['config']['adam'] == 'one launch after the backward', d['config']['adam']
    assert abs(d['value'] - ranks * 1000.0 / d['ms_per_step']) / d['value'] < 0.01  # whole-job: every rank's view per step
    assert 'cpu_baseline' not in d


@pytest.mark.parametrize('port,ranks', [(29591, 2), (29592, 4)])
def test_default_multi_rank_run_ranks_every_exchange_variant(port, ranks):
    """`bench.py --gpus N` with NO exchange flag (what the driver's scaling run issues): every exchange variant is timed in
    the same processes, each must keep the replicas bit-identical, the reported value is the fastest variant's, and all
    variants trained the same parameters up to summation order"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '6',
           '--warmup', '2']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    ev = d['exchange_variants']
    assert set(ev) == {'allreduce', 'factors', 'factors-overlap', 'pipeline', 'allreduce-graph', 'factors-graph',
                       'factors-graph-split'}, ev
    # the two variants with the collectives captured inside the step graph need RCCL: over gloo (this test shares one GPU
    # between the ranks) they must fail cleanly on every rank and cost nothing but their entry
    for name in ('allreduce-graph', 'factors-graph', 'factors-graph-split'):
        assert 'RCCL backend' in ev.pop(name)['error']
    for name, r in ev.items():
        assert 'error' not in r and r['replicas_identical'] is True, (name, r)
    best = d['config']['exchange']
    assert best == max(ev, key=lambda n: ev[n]['value']) and d['value'] == ev[best]['value']
    assert d['n_gpus'] == ranks and d['steps'] == 6 and d['config']['replicas_identical'] is True
    assert abs(d['value'] - ranks * 1000.0 / d['ms_per_step']) / d['value'] < 0.01
    digests = [r['param_digest'] for r in ev.values()]
    assert max(digests) - min(digests) <= 1e-6 * abs(digests[0]), digests
    assert d['ms_per_step_blocks']['blocks'] == 6 and 'cpu_baseline' not in d
    prov = [l for l in p.stderr.splitlines() if '[exchange auto] provisional {' in l]
    assert len(prov) == 4 and "after variant 'allreduce': 1 of 7" in prov[0], [l[:200] for l in prov]   # (one per variant that ran)


def test_factor_exchange_and_dense_exchange_train_the_same_parameters():
    """two ranks, eight optimizer steps: the SH gradient exchanged as factors (--sh-factors) or all-reduced densely
    (default) must leave the same parameters up to summation order"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    digests = []
    for port, extra in ((29585, ['--sh-factors']), (29586, PLAIN)):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
               '--warmup', '2', '--no-cpu-baseline', '--eager'] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][0])
        assert d['config']['replicas_identical'] is True
        digests.append(d['config']['param_digest'])
    assert abs(digests[0] - digests[1]) <= 1e-7 * abs(digests[1]), digests


def test_riding_update_and_plain_update_train_the_same_parameters():
    """two ranks, eight optimizer steps: the update as closing launch + next view's skeleton forward with the rows' Adam on
    board (default) or as one Adam launch (--pre-forward off) must leave the same parameters (same arithmetic per element;
    the blend backward's atomics do not order their additions, hence a tolerance)"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    digests = []
    for port, extra in ((29589, PLAIN), (29590, PLAIN + ['--pre-forward', 'off'])):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
               '--warmup', '2', '--no-cpu-baseline'] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][0])
        assert d['config']['replicas_identical'] is True
        digests.append(d['config']['param_digest'])
    assert abs(digests[0] - digests[1]) <= 1e-7 * abs(digests[1]), digests


def test_two_rank_training_with_densification_keeps_the_replicas_identical():
    """examples/train_views.py on two ranks (gloo, one GPU): view-parallel steps, densification statistics all-reduced
    (SUM, SUM, MAX) before every clone / split / prune so the ranks take identical decisions, the overflow guard in the
    loop; at the end the ranks compare a digest of every parameter"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29588', os.path.join(ROOT, 'examples', 'train_views.py'), '--iters', '130', '--densify-every', '50',
           '--gaussians', '8000', '--size', '160']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert p.stdout.count('densify:') == 2, p.stdout[-1500:]
    assert 'replicas identical: True' in p.stdout, p.stdout[-1500:]


@pytest.mark.parametrize('port,exchange', [(29597, 'allreduce-graph'), (29598, 'factors-graph'), (29599, 'factors-graph-split')])
def test_collectives_captured_inside_the_step_graph_one_rank_rccl(port, exchange):
    """`--exchange allreduce-graph / factors-graph`: the RCCL collectives are nodes of the ONE step graph (the factor
    all-gather as a branch beside the skinning backward; `-split`: the rows' all-reduce beside the skeleton backward).  One GPU holds one RCCL rank, so this runs the real backend with a
    1-rank group: capture, replays, the update path of the multi-rank step; the result must train like the eager-collective
    form of the same exchange"""
    env = dict(os.environ, SKGS_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = {}
    eager_form = {'allreduce-graph': 'allreduce', 'factors-graph': 'factors-overlap', 'factors-graph-split': 'factors-overlap'}
    for k, x in enumerate((exchange, eager_form[exchange])):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
               '--master-port', str(port + 10 * k), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '12', '--warmup', '3',
               '--no-cpu-baseline', '--no-ms-per-render', '--exchange', x]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
        out[x] = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][-1])
    a, b = out.values()
    assert a['config']['replicas_identical'] is True and b['config']['replicas_identical'] is True
    assert abs(a['config']['param_digest'] - b['config']['param_digest']) <= 1e-6 * abs(b['config']['param_digest'])
    assert 'ONE captured hipGraph' in a['config']['launch']


def test_auto_ranking_stops_starting_variants_when_its_time_budget_is_spent():
    """`--auto-budget` (seconds): the record is one line at the very end, so the ranking must end by itself before a caller's
    limit could cut it off -- with a budget of zero only the first variant runs (the plain flat-buffer
    all-reduce: one standard collective per step), the others are listed as not run"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29641', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--auto-budget', '0']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][-1])
    ev = d['exchange_variants']
    assert list(ev)[0] == 'allreduce' and d['config']['exchange'] == 'allreduce' and 'value' in ev['allreduce']
    for name, r in ev.items():
        if name != 'allreduce':
            assert 'budget' in r['error'], (name, r)
    # the record of the first finished variant is on stderr BEFORE the ranking ends (a lease that dies later keeps it)
    prov = [l for l in p.stderr.splitlines() if '[exchange auto] provisional {' in l]
    assert prov, p.stderr[-2000:]
    pj = json.loads(prov[0].split('provisional ', 1)[1])
    assert pj['n_gpus'] == 2 and pj['config']['exchange'] == 'allreduce' and pj['value'] == ev['allreduce']['value'] and 'provisional' in pj
    # the record proves the ranks: backend, world, one entry per rank with its device
    cl = d['config']['cluster']
    assert cl['world'] == 2 and cl['backend'] == 'gloo' and len(cl['devices']) == 2 and {x['rank'] for x in cl['devices']} == {0, 1}


def test_gpus_flag_without_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (VERDICT r3 #5a): the ranks are spawned (here: sharing this one GPU
    over gloo), rank 0's line says n_gpus = 2"""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                        '--exchange', 'allreduce', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['config']['cluster']['world'] == 2 and d['config']['replicas_identical'] is True


def test_stage_sp_prints_the_contract_line():
    """`bench.py --stage sp`: the superpoint stage at config #1's size -- the same contract fields, the stage's own
    kernels in the table (the MFMA network against the fp32 matrix peak, the search against its algorithmic bytes)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--stage', 'sp', '--steps', '8', '--warmup', '2',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'kernels'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['config']['stage'] == 'sp' and '512 superpoints' in d['config']['workload']
    assert abs(d['value'] - 1000.0 / d['ms_per_step']) / d['value'] < 0.01
    k = d['kernels']
    # (the skinning and the rows pass of its backward are jobs of the rasterizer's per-Gaussian launches)
    assert 'deform_forward' in k['preprocess_forward']['includes'] and 'rows pass' in k['preprocess_backward']['includes']
    for name in ('sp_net_forward', 'sp_net_backward', 'sp_knn_weights', 'preprocess_forward', 'preprocess_backward', 'deform_backward',
                 'render_forward', 'render_backward', 'adam'):
        assert k[name]['us'] > 0 and 0 < k[name]['frac'] < 1, (name, k[name])
    assert 0 < k['sp_net_forward']['frac_of_mfma_f32_peak'] < 1 and k['sp_net_forward']['TFLOPs'] > 1
    total = sum(v['us'] * v['launches_per_step'] for v in k.values())
    assert 0.8 * d['ms_per_step'] * 1e3 < total < 1.6 * d['ms_per_step'] * 1e3
    assert d['ms_per_step'] < 1.0  # (0.49 ms when written: within 1.5x of the skeleton stage's step)
    assert d['fps_forward_render']['fused_step_graph'] > 500  # (the reference's FPS protocol on the stage's forward: ~5500)


def test_stage_sp_two_ranks_share_the_gpu_and_stay_identical():
    """the superpoint stage view-parallel: graph(forward + backward) | ONE all-reduce of the flat gradient buffer | graph(Adam);
    two ranks on this one GPU over gloo must end with bit-identical replicas"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29597', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stage', 'sp', '--steps', '6', '--warmup', '2',
           '--no-cpu-baseline']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['replicas_identical'] is True and d['config']['exchange'] == 'allreduce'
    assert d['config']['cluster']['world'] == 2
    assert abs(d['value'] - 2 * 1000.0 / d['ms_per_step']) / d['value'] < 0.01


def test_the_driver_command_carries_the_baselines_and_the_reference_route():
    """`python bench.py` with NO flags -- the driver's round-end command: one JSON line within minutes, with `cpu_baseline`, `survey_recipe` and
    (round 6) `reference_route`: the reference's own iteration on the hooks as child runs, every iteration of the fused ones on the route"""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')], capture_output=True, text=True, timeout=900, cwd=ROOT)
    took = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert took < 420, took
    assert d['n_gpus'] == 1 and d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0 and d['survey_recipe']['value'] > 0
    rr = d['reference_route']
    for k in ('sk_accelerated', 'sk_fused', 'sk_fused_headline_scene', 'sp_fused'):
        assert 'error' not in rr[k], rr[k]
        assert rr[k]['value'] > 0 and rr[k]['unit'] == 'iters/s' and rr[k]['how'].startswith('python bench.py --config 1')
    for k in ('sk_fused', 'sk_fused_headline_scene', 'sp_fused'):
        r = rr[k]['route']
        assert r['render_reference'] == 0 and r['render_fused'] == rr[k]['steps'] + rr[k]['prime_steps'] + 10 and r['status']['overflow_events'] == 0
    assert rr['sk_fused']['value'] > 2 * rr['sk_accelerated']['value']
    assert d['value'] > rr['sk_fused_headline_scene']['value'] > 0.6 * d['value']          # (the package's own step: no Python between the launches)


@pytest.mark.parametrize('stage', ['sk', 'sp'])
def test_reference_loop_runs_on_the_hooks_and_accelerated_with_the_same_training(stage):
    """`bench.py --reference-loop hooks | accelerated | fused` (benchlib/reference_loop.py): the reference's whole iteration restated on the hooks
    alone and after accelerate_reference() -- one JSON line each, every accelerator used on every step of the second run, the same loss
    after the same steps (different arithmetic paths, same training), and the accelerated loop faster"""
    out = {}
    for mode in ('hooks', 'accelerated', 'fused'):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--stage', stage, '--reference-loop', mode, '--steps', '12', '--warmup', '5',
                            '--prime-steps', '0'], capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        out[mode] = json.loads(lines[0])
    h, a = out['hooks'], out['accelerated']
    assert h['config']['accelerators'] is None and h['config']['lie_fused_calls']['forward'] >= 17     # the recognised skinning expression
    c = a['config']['accelerators']
    for k in ('ssim', 'lbs_weight') + (('kinematic', 'sk_net') if stage == 'sk' else ('sp_net',)):
        assert c[f'{k}_fused'] >= 17 and c[f'{k}_reference'] == 0, (k, c)
    assert c['adam_fused'] >= 16 and c['adam_reference'] == 1           # (the first step creates torch's state)
    assert abs(h['config']['loss_last'] - a['config']['loss_last']) <= 2e-3 * abs(h['config']['loss_last'])
    assert a['value'] > 1.5 * h['value']
    # the same loop with SkeletonGaussianSplatting.render + the two image-loss classes routed into the fused step
    # (sk_gs_amd/reference_fused.py): every iteration on the route, the plain backward graph, no overflow, the same loss, and faster again
    f = out['fused']
    fr = f['config']['fused_route']
    assert fr['calls']['render_fused'] == 17 and fr['calls']['render_reference'] == 0 and fr['calls']['backward_direct'] == 17, fr
    assert fr['calls']['image_terms_fused'] == 17 and fr['calls']['foreign_grads_added'] == 0 and fr['calls']['backward_extras'] == 0
    assert fr['status']['overflow_events'] == 0 and fr['status']['tile_bucket'] >= 64
    assert f['config']['accelerators']['adam_fused'] >= 15
    assert abs(f['config']['loss_last'] - a['config']['loss_last']) <= 2e-3 * abs(a['config']['loss_last'])
    assert f['value'] > 1.5 * a['value'] and f['prime_steps'] == 0
