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


def pytest_sessionfinish(session, exitstatus):
    """what the robust comparisons observed (outlier fraction, max error per tensor) -> gpurun_out/parity_observed.json,
    so the numbers behind the tolerances of tests/helpers.assert_close_robust are kept, not only asserted"""
    import json
    try:
        import helpers
    except Exception:
        return
    if not helpers.OBSERVED:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_observed.json'), 'w') as f:
            json.dump(helpers.OBSERVED, f, indent=1)
    except OSError:
        pass
