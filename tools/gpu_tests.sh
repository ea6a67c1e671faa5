#!/bin/bash
# The GPU suite N times with NOTHING swallowed: the whole log is kept (tee), pytest's exit code is recorded per run (pipefail),
# every run's parity_observed.json is kept (five of them refresh the drift gate's baseline: tools/update_parity_baseline.py).
# usage (GPU box, repo root): bash tools/gpu_tests.sh <tag> [runs=1]   ->  gpurun_out/<tag>/gpu_tests_run<i>.txt, gpu_tests_rc.txt
# exit code: 0 only if every run's was 0.          (round 5's script piped pytest through `tail -3` and shipped a red gate unseen)
set -o pipefail
tag=${1:-rXX}; runs=${2:-1}
out=gpurun_out/$tag; mkdir -p "$out"
: > $out/gpu_tests_rc.txt
worst=0
for i in $(seq 1 $runs); do
  python -m pytest tests -m gpu -q 2>&1 | tee $out/gpu_tests_run$i.txt | tail -4
  rc=$?
  echo "run $i: rc=$rc $(grep -E '^[0-9]+ (passed|failed)|passed|failed' $out/gpu_tests_run$i.txt | tail -1) | $(grep -F '[parity] gate:' $out/gpu_tests_run$i.txt | tail -1)" | tee -a $out/gpu_tests_rc.txt
  cp gpurun_out/parity_observed.json $out/parity_observed_run$i.json 2>/dev/null
  [ $rc -ne 0 ] && worst=$rc
done
cp $out/parity_observed_run1.json $out/parity_observed.json 2>/dev/null
cp $out/gpu_tests_run$runs.txt $out/gpu_tests.txt
echo "gpu tests: $runs run(s), worst rc=$worst" | tee -a $out/gpu_tests_rc.txt
exit $worst
