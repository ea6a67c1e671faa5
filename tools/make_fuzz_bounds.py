#!/usr/bin/env python3
"""tests/golden/fuzz_bounds.json from a GPU session's parity_observed.json: every tensor of tests/test_gpu_fuzz.py whose worst untraced row is
more than 1e-4 from the fp32 oracle (it passes through the second branch of tests/helpers.FlipCensus.check_rows), with the reference
arithmetic's own error on it (ref_err = |fp32 oracle - fp64 oracle|), the product's two errors and the bound it was held to; plus the
distribution of (product error vs fp64) / ref_err over all tensors with ref_err > 3e-5.
usage: python tools/make_fuzz_bounds.py [gpurun_out/parity_observed.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_observed.json')
rows = [r for r in json.load(open(src)) if 'ref_err' in r and 'test_gpu_fuzz' in r.get('test', '')]
need, ratios = [], []
for r in rows:
    seed = int(r['test'].split('[')[-1].rstrip(']'))
    tensor = r['name'].split(' ')[2]
    if r['ref_err'] > 3e-5:
        ratios.append(r['untraced_max_vs_fp64'] / r['ref_err'])
    if r['untraced_max'] > r['tol']:
        need.append(dict(seed=seed, tensor=tensor, ref_err=r['ref_err'], err_vs_fp32_oracle=r['untraced_max'],
                         err_vs_fp64_oracle=r['untraced_max_vs_fp64'], bound_vs_fp64=max(r['tol'], helpers.REF_ERR_FACTOR * r['ref_err']),
                         ratio=r['untraced_max_vs_fp64'] / r['ref_err']))
ratios.sort()
out = dict(factor=helpers.REF_ERR_FACTOR, tensors_checked=len(rows), tensors_with_ref_err_over_3e5=len(ratios),
           ratio_product_to_reference_error=dict(min=ratios[0], median=ratios[len(ratios) // 2], p90=ratios[int(0.9 * len(ratios))], max=ratios[-1]),
           second_branch=sorted(need, key=lambda e: (e['seed'], e['tensor'])))
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'fuzz_bounds.json'), 'w'), indent=1)
print(json.dumps(out['ratio_product_to_reference_error']), len(need), 'tensors on the second branch')
for e in out['second_branch']:
    print(f"  seed {e['seed']:5d} {e['tensor']:14s} ref_err {e['ref_err']:.2e}  vs fp32 {e['err_vs_fp32_oracle']:.2e}  vs fp64 {e['err_vs_fp64_oracle']:.2e}  "
          f"bound {e['bound_vs_fp64']:.2e}  ratio {e['ratio']:.2f}")
