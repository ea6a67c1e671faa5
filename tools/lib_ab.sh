#!/bin/bash
# alternate bench.py between the in-tree library and a variant build (SKGS_HIP_LIB): usage: bash tools/lib_ab.sh <variant.so> [reps=4] [bench args...]
var=$1; reps=${2:-4}; shift; shift
for rep in $(seq 1 $reps); do
  for which in base variant; do
    if [ $which = variant ]; then export SKGS_HIP_LIB=$var; else unset SKGS_HIP_LIB; fi
    python bench.py --steps 400 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$which rep $rep', d['value'], d['ms_per_step'], {n:k[n]['us'] for n in k if 'skeleton' in n or 'preprocess' in n})"
  done
done
