#!/bin/bash
# A/B of an environment knob inside one box: scratch/env_ab.sh VAR val1 val2 ...
VAR=$1; shift
for round in 1 2; do
for v in "$@"; do
  echo "== $VAR=$v (round $round)"
  env $VAR=$v python scratch/perf_full.py 32 2>&1 | grep -E "iter 2"
done
done
