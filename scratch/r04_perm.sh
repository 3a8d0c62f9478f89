#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_decoder.py tests/test_gpu_full.py -x -q 2>&1 | tail -4 > gpurun_out/r4_perm_test.log
rc=$?
cat gpurun_out/r4_perm_test.log
if [ $rc -ne 0 ]; then exit 1; fi
for b in 32 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 qa= off=qa:0 > gpurun_out/r4_perm_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_perm_b$b.txt; exit 1; }
  tail -3 gpurun_out/r4_perm_b$b.txt
done
