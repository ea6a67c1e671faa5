import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle32():
    from oracle.oracle import Oracle, build
    build()
    return Oracle('f32')


@pytest.fixture(scope='session')
def oracle64():
    from oracle.oracle import Oracle, build
    build()
    return Oracle('f64')


def _regressions(observed):
    """tests/parity_gate.py: census entries whose worst case over the elements NO flip explains left the band the committed
    baseline (max / min over >= 5 GPU sessions per entry) allows, or whose flipped pixels more than doubled.  Comparisons of
    trained parameters (atomics' order, Adam's sign steps) vary run to run and are held by their own assertions only."""
    import parity_gate
    return parity_gate.regressions(observed, parity_gate.load_baseline())


def _print_baseline_log(last=3):
    """the tail of tests/golden/parity_baseline_log.json: every refresh of the baseline the 2x gate compares against (the one
    way to loosen it) with its stated reason and the entries that got looser -- printed by every session that gates"""
    import json
    path = os.path.join(ROOT, 'tests', 'golden', 'parity_baseline_log.json')
    if not os.path.exists(path):
        return
    try:
        log = json.load(open(path))
    except (OSError, ValueError):
        return
    print(f'\n[parity] baseline refreshes on record: {len(log)}; latest:')
    for e in log[-last:]:
        print(f"  {e.get('date')} ({e.get('commit') or 'no commit given'}): {e.get('reason')} -- {e.get('refreshed')} entries, "
              f"{e.get('looser_count', 0)} looser, {e.get('tighter', 0)} tighter")
        for l in e.get('looser', [])[:3]:
            print(f"      looser: {l['test']} [{l['name']}] {l['field']}: {l['old']:.3g} -> {l['new']:.3g}")


def pytest_sessionfinish(session, exitstatus):
    """what the comparisons observed (outlier fraction, max error, flips per tensor) -> gpurun_out/parity_observed.json, so
    the numbers behind the tolerances of tests/helpers are kept, not only asserted; and the session FAILS if a worst case
    left the band of tests/golden/parity_observed_baseline.json (tests/parity_gate.py; refresh with
    tools/update_parity_baseline.py from >= 5 sessions after a deliberate numerics change, and say why).  The verdict line
    `[parity] gate: ...` is always printed and also written to gpurun_out/parity_gate.txt."""
    import json
    try:
        import helpers
    except Exception:
        return
    if not helpers.OBSERVED:
        return
    try:
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        # (a session without a GPU -- the oracle's own census tests -- must not overwrite the file of the last GPU session)
        import torch
        name = 'parity_observed.json' if torch.cuda.is_available() else 'parity_observed_cpu.json'
        with open(os.path.join(out, name), 'w') as f:
            json.dump(helpers.OBSERVED, f, indent=1)
    except OSError:
        pass
    _print_baseline_log()
    bad = _regressions(helpers.OBSERVED)
    gated = sum(1 for r in helpers.OBSERVED if 'untraced_max' in r)
    verdict = (f'[parity] gate: RED -- {len(bad)} of {gated} census entries left their baseline band' if bad else
               f'[parity] gate: green -- {gated} census entries inside their baseline band')
    print('\n' + verdict + ('\n  ' + '\n  '.join(bad) if bad else ''))
    try:
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_gate.txt' if torch.cuda.is_available() else 'parity_gate_cpu.txt'), 'w') as f:
            f.write(verdict + '\n' + '\n'.join(bad) + ('\n' if bad else ''))
    except (OSError, NameError):
        pass
    if bad:
        session.exitstatus = 1
