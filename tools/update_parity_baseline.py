"""Refresh tests/golden/parity_observed_baseline.json from the last GPU test run (gpurun_out/parity_observed.json).

tests/conftest.py fails a session whose worst observed error grew more than 2x over this baseline; run this after a
DELIBERATE numerics change and state the reason in the commit message.  Keeps, per (test, tensor), the largest error of
the run; entries of tests that did not run keep their old value."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_observed.json')
dst = os.path.join(ROOT, 'tests', 'golden', 'parity_observed_baseline.json')
old = {}
if os.path.exists(dst):
    old = {(r['test'], r['name']): r for r in json.load(open(dst)) if 'untraced_max' in r}
new = {}
for r in json.load(open(src)):
    if 'untraced_max' not in r:  # only the deterministic census comparisons are gated (tests/conftest.py::_regressions)
        continue
    k = (r.get('test', ''), r['name'])
    keep = {f: r[f] for f in ('test', 'name', 'elements', 'tol', 'max_err', 'frac_over_tol') if f in r}
    for f in ('flipped_pixels', 'flipped_max_margin', 'traced_pixels', 'traced_rows', 'rows_touching_a_flip', 'untraced_max'):
        if f in r:
            keep[f] = r[f]
    if k not in new or keep['max_err'] > new[k]['max_err']:
        new[k] = keep
old.update(new)
rows = sorted(old.values(), key=lambda r: (r['test'], r['name']))
json.dump(rows, open(dst, 'w'), indent=0)
print(f'{len(new)} entries refreshed, {len(rows)} in {dst}')
