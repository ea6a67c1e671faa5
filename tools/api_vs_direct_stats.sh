#!/bin/bash
# the statistics behind test_fused_train_step_behind_the_autograd_api's thresholds: N runs, worst loss difference and far fraction
n=${1:-40}
for i in $(seq 1 $n); do
  python -m pytest tests/test_gpu_fused_step.py::test_fused_train_step_behind_the_autograd_api -q -s 2>&1 | grep "api-vs-direct\|passed\|failed" > /tmp/avd.txt
  python - <<P
import re
L=open('/tmp/avd.txt').read().splitlines()
loss=[l for l in L if 'losses' in l]
far=[(float(re.search(r'far ([0-9.e+-]+)', l).group(1)), l.split()[1]) for l in L if 'far' in l]
print('run $i', loss[0].split(':')[1].strip() if loss else '-', 'worst far', max(far) if far else '-', L[-1] if L else '')
P
done
