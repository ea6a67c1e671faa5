#!/bin/bash
# A/B of the blend kernels' tile order: default bench + rocprofv3 kernel stats with the groups heaviest first
# (SKGS_TILE_ORDER=1, the default) and in raster order (0).  usage on the GPU box: bash tools/ab_tile_order.sh <tag> [bench args]
tag=${1:-r03_order}
EXTRA="${@:2}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
for mode in 1 0 1 0; do
  SKGS_TILE_ORDER=$mode python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('order=$mode', d['value'], 'it/s', d['ms_per_step'], 'ms', d['ms_per_step_blocks'], 'fwd', k['render_forward']['us'], 'bwd', k['render_backward']['us'], 'sort', k['tile_sort']['us'])"
done > $out/ab.txt 2>&1
for mode in 1 0; do
  export SKGS_TILE_ORDER=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$mode -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-ms-per-render $EXTRA > $out/stats$mode.log 2>&1
  python - <<PY >> $out/ab.txt
import csv,glob
f=glob.glob('$out/stats$mode/*/*kernel_stats.csv')[0]
print('--- rocprofv3 kernel stats, SKGS_TILE_ORDER=$mode')
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:70]:<70} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
  cp $(ls $out/stats$mode/*/*kernel_stats.csv | head -1) $out/kernel_stats_order$mode.csv
done
find $out -name '*kernel_trace.csv' -delete
cat $out/ab.txt
