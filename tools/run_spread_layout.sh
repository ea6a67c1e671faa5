#!/bin/bash
# tools/run_spread.sh with the optimizer arrays' addresses printed beside each run's time (SKGS_PRINT_LAYOUT)
n=${1:-12}
for i in $(seq 1 $n); do
  SKGS_PRINT_LAYOUT=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/tmp/lay.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('run $i', d['ms_per_step'], k['skeleton_backward']['us'], end=' ')"
  grep "\[layout\]" /tmp/lay.err | cut -c60-420
done
