#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for env in "" "QASR_DA_UNR=1" "QASR_DA_UNR=1 QASR_GEMV_W1024=4"; do
  echo "== env: $env" >> gpurun_out/r4_lanes.txt
  env $env timeout -k 10 400 python scratch/lanes_sweep.py --lanes 1,3 --steps 9 >> gpurun_out/r4_lanes.txt 2>&1 || { tail -5 gpurun_out/r4_lanes.txt; exit 1; }
done
grep -v amdgpu gpurun_out/r4_lanes.txt | tail -12
