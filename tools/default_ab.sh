#!/bin/bash
# the driver's command (python bench.py, no flags) under SKGS_MLP_XCD=0 / 1, alternating, beside the short form the config lines use
reps=${1:-2}
for rep in $(seq 1 $reps); do
  for m in 0 1; do
    SKGS_MLP_XCD=$m python bench.py --no-reference-route 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default      mode $m rep $rep', d['value'], d['ms_per_step'], d['ms_per_step_blocks']['min'], d['ms_per_step_blocks']['max'])"
    SKGS_MLP_XCD=$m python bench.py --config 1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1 x100 mode $m rep $rep', d['value'], d['ms_per_step'], d['ms_per_step_blocks']['min'], d['ms_per_step_blocks']['max'])"
    SKGS_MLP_XCD=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x200 no-cpu  mode $m rep $rep', d['value'], d['ms_per_step'], d['ms_per_step_blocks']['min'], d['ms_per_step_blocks']['max'])"
  done
done
