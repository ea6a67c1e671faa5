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
    assert abs(d['value'] - 1000.0 / d['ms_per_step']) / d['value'] < 0.01


@pytest.mark.parametrize('port,ranks,extra', [(29577, 2, []), (29578, 2, ['--pipeline']), (29579, 2, ['--dense-spw-grad']),
                                              (29580, 2, ['--autograd']), (29581, 4, []), (29582, 2, ['--sh-allreduce']), (29583, 2, ['--overlap-gather']), (29584, 2, ['--deform-net'])])
def test_ranks_share_the_gpu_and_stay_identical(port, ranks, extra):
    """The driver's multi-GPU launch line with 2 or 4 ranks on this one GPU (gloo moves the gradients: RCCL refuses two ranks
    per device): the real view-parallel schedule -- graph(fwd+bwd) | all-reduce | graph(expand + Adam) -- must keep the
    replicas bit-identical and print one line from rank 0 with whole-job throughput."""
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
    assert abs(d['value'] - ranks * 1000.0 / d['ms_per_step']) / d['value'] < 0.01  # whole-job: every rank's view per step
    assert 'cpu_baseline' not in d


def test_factor_exchange_and_dense_exchange_train_the_same_parameters():
    """two ranks, eight optimizer steps: the SH gradient exchanged as factors (default) or all-reduced densely
    (--sh-allreduce) must leave the same parameters up to summation order"""
    env = dict(os.environ, SKGS_DIST_BACKEND='gloo', SKGS_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    digests = []
    for port, extra in ((29585, []), (29586, ['--sh-allreduce'])):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
               '--warmup', '2', '--no-cpu-baseline', '--eager'] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][0])
        assert d['config']['replicas_identical'] is True
        digests.append(d['config']['param_digest'])
    assert abs(digests[0] - digests[1]) <= 1e-7 * abs(digests[1]), digests
