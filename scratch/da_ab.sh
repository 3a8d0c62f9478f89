#!/bin/bash
# A/B of the decode attention variants: probe + stamps + in-situ decode time
for cfg in "0 16" "1 16" "2 16" "1 8" "2 8"; do
  set -- $cfg
  for b in 32 1; do
    echo "== MFMA=$1 WAVES=$2 B=$b"
    QASR_DA_MFMA=$1 QASR_DA_WAVES=$2 QASR_DA_STAMPS=1 python scratch/probe_ab.py $b 2>&1 | grep -E "stamps|round 2 probe decode|decode" | tail -4
  done
done
