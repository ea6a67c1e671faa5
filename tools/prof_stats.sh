cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_s; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/prof_s.log 2>&1; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_s/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:20]:
    print(f"{r['Name'][:70]:<70} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
