cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03g
python -m pytest tests/test_gpu_raster.py tests/test_gpu_fullsize.py tests/test_gpu_variants.py tests/test_gpu_fused_step.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r03g/gputest.txt; cat gpurun_out/r03g/gputest.txt
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(d['value'], 'it/s', d['ms_per_step'], 'ms fwd', k['render_forward']['us'], 'bwd', k['render_backward']['us'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g/stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-ms-per-render > gpurun_out/r03g/stats.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03g/stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:15]:
    print(f"{r['Name'][:70]:<70} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
find gpurun_out/r03g -name '*kernel_trace.csv' -delete
