"""The hand-placed cross-lane instruction sequences are checked on the hardware they were written for: the probe of the
nine-value wave reduction (bank-masked DPP adds, v_permlane32/16_swap) is built and run."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_reduction_probe(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    exe = tmp_path / 'permlane_probe'
    b = subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', '-I', os.path.join(ROOT, 'sk_gs_amd', 'csrc'),
                        os.path.join(ROOT, 'tools', 'permlane_probe.hip'), '-o', str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'PROBE OK' in r.stdout, r.stdout[-2000:]
    assert r.stdout.count('banked lane') == 9
