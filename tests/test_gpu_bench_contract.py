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
