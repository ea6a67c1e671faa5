# kernel statistics of the operator-path training step (bench.py --autograd): which launches a step is made of
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_ag
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ag -- python bench.py --autograd --steps 200 --warmup 10 --prime-steps 0 --no-cpu-baseline --no-ms-per-render < /dev/null > gpurun_out/prof_ag.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_ag/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
tail -1 gpurun_out/prof_ag.log | cut -c1-300
