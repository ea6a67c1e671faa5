#!/bin/bash
# run one test N times per library build; print every failure's assertion lines.  usage: bash tools/flake_hunt.sh <pytest node id> [N=40] [variant.so]
node=$1; n=${2:-40}; var=$3
for which in base variant; do
  [ $which = variant ] && [ -z "$var" ] && continue
  if [ $which = variant ]; then export SKGS_HIP_LIB=$var; else unset SKGS_HIP_LIB; fi
  fails=0
  for i in $(seq 1 $n); do
    python -m pytest $node -x -q > /tmp/flake.txt 2>&1 || { fails=$((fails+1)); echo "--- $which run $i"; grep -E "^E  |^tests/.*(Error|assert)|^>" /tmp/flake.txt | head -12; }
  done
  echo "$which: $fails of $n failed"
done
