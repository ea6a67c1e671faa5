#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command (run on the GPU box): tools/prof_cmd.sh <tag> python3 script.py args...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_$tag/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:25]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
