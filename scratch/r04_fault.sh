#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -8 > gpurun_out/r4_fault_test.log
rc=$?
cat gpurun_out/r4_fault_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scratch/knob_ab.py --batch 32 --rounds 5 qa= off=qa:0 > gpurun_out/r4_after_fault_ab.txt 2>&1 || { tail -5 gpurun_out/r4_after_fault_ab.txt; exit 1; }
tail -3 gpurun_out/r4_after_fault_ab.txt
