#!/bin/bash
# A/B of where the deform network's 32 workgroups run (skgs_deform_mlp_xcd_mode / SKGS_MLP_XCD): 0 = blocks 0..31, write-through exchange;
# 1 = one XCD, exchange in its L2 after the launch's own census (default); 2 = that placement, write-through; 3 = mode 1, falsified census
# usage (GPU box, repo root): bash tools/xcd_ab.sh [tag=xcd] [reps=3]   ->  gpurun_out/<tag>/
tag=${1:-xcd}; reps=${2:-3}
out=gpurun_out/$tag; mkdir -p $out
for m in 0 1 2 3; do
  SKGS_MLP_XCD=$m python tools/time_mlp.py > $out/time_mlp_$m.txt 2>&1
  SKGS_MLP_XCD=$m python tools/time_skeleton.py > $out/time_skeleton_$m.txt 2>&1
done
: > $out/summary.txt
for rep in $(seq 1 $reps); do
  for m in 0 1 2 3; do
    SKGS_MLP_XCD=$m python bench.py --steps 400 --no-cpu-baseline > $out/bench_${m}_$rep.json 2> $out/bench_${m}_$rep.err
    python - <<P | tee -a $out/summary.txt
import json
d=json.loads(open('$out/bench_${m}_$rep.json').read().strip().splitlines()[-1])
k=d['kernels']
print('mode $m rep $rep', d['value'], d['ms_per_step'], {n:k[n]['us'] for n in k if 'skeleton' in n or 'mlp' in n}, d['config'].get('deform_net_on_one_xcd', {}).get('forward'))
P
  done
done
for m in 0 1; do
  SKGS_MLP_XCD=$m python bench.py --stage sp --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sp mode $m', d['value'], d['ms_per_step'])" | tee -a $out/summary.txt
  SKGS_MLP_XCD=$m python bench.py --config 4 --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4 mode $m', d['value'], d['ms_per_step'])" | tee -a $out/summary.txt
  SKGS_MLP_XCD=$m python bench.py --reference-loop fused --steps 200 --loop-scene headline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reference-loop fused headline scene mode $m', d['value'], d['ms_per_step'])" | tee -a $out/summary.txt
done
