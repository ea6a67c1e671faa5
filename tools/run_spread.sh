#!/bin/bash
# the same bench command N times in fresh processes on one box: how far apart are runs of one build?  usage: bash tools/run_spread.sh [N=12] [bench args...]
n=${1:-12}; shift
for i in $(seq 1 $n); do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('run $i', d['ms_per_step'], d['ms_per_step_blocks']['min'], d['ms_per_step_blocks']['max'], {n:k[n]['us'] for n in ('skeleton_forward','skeleton_backward','adam','render_backward')}, (d['config'].get('deform_net_on_one_xcd') or {}).get('xcds'))"
done
