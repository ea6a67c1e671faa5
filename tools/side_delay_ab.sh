#!/bin/bash
# experiment: the side job's workgroups of the skeleton launches start late (SKGS_SIDE_DELAY x ~0.85 us)
for rep in 1 2; do
for d in 0 2 4 6 9; do
  SKGS_SIDE_DELAY=$d python bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('delay $d rep $rep', d['value'], d['ms_per_step'], {n:k[n]['us'] for n in k if 'skeleton' in n})"
done
done
for d in 0 4 9; do echo "delay $d"; SKGS_SIDE_DELAY=$d python tools/time_skeleton.py 2>&1 | grep -v "^P=" | tail -9; done
