#!/bin/bash
for cfg in "0 16" "1 16" "1 8" "2 8"; do
  set -- $cfg
  echo "== MFMA=$1 WAVES=$2"
  QASR_DA_MFMA=$1 QASR_DA_WAVES=$2 python scratch/perf_full.py 32 2>&1 | grep -E "iter 2|probe decode"
done
