cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_a; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --autograd > gpurun_out/prof_a.log 2>&1; tail -1 gpurun_out/prof_a.log | cut -c1-300; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_a/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:45]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
