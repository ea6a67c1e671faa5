#!/bin/bash
# One gpurun call: kernel stats + three separate PMC passes of bench.py (never combined with other trace domains).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
tag=${1:-rXX}
EXTRA="${@:2}"   # extra bench.py arguments, e.g. --config 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline $EXTRA > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $out/write.log 2>&1
rocprofv3 --pmc VALUBusy VALUUtilization SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/valu -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $out/valu.log 2>&1
grep -h '^{"metric"' $out/stats.log | tail -1 > $out/bench_steps100.json
# the trace csv files are large: keep the summaries
find $out -name '*kernel_trace.csv' -delete
ls -R $out | head -40
