#!/bin/bash
# does the placement of the deform network change what is trained?  bench.py's timed steps (graph replays, 4 steps each) under
# SKGS_MLP_XCD=0 and =1, twice each; per parameter tensor the relative difference of sum and norm between runs: 0 vs 0' is the
# noise of the float atomics, 0 vs 1 must be no larger.  usage (GPU box): bash tools/mode_equivalence.sh [steps=200]
steps=${1:-200}
for tag in a0 b0 a1 b1; do
  m=${tag:1:1}
  SKGS_MLP_XCD=$m SKGS_PRINT_DIGEST=1 python bench.py --steps $steps --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep "^\[digest\]" | cut -c10- > /tmp/digest_$tag.json
done
python - <<P
import json
D={t:json.load(open(f'/tmp/digest_{t}.json')) for t in ('a0','b0','a1','b1')}
def rel(x,y):
    return max(abs(x[n][k]-y[n][k])/max(abs(y[n][k]),1e-30) for n in x for k in (0,1)), max(((abs(x[n][1]-y[n][1])/max(abs(y[n][1]),1e-30)), n) for n in x)
for a,b in (('a0','b0'),('a1','b1'),('a0','a1'),('b0','b1'),('a0','b1')):
    r=rel(D[a],D[b]); print(f'{a} vs {b}: worst relative difference of a tensor\'s sum / norm {r[0]:.2e}; worst norm: {r[1][0]:.2e} ({r[1][1]})')
P
