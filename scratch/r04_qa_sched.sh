#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -4 > gpurun_out/r4_qa_test.log
rc=$?
cat gpurun_out/r4_qa_test.log
if [ $rc -ne 0 ]; then exit 1; fi
for b in 32 8 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 e1= e3=qa_early:3 e4=qa_early:4 off=qa:0 > gpurun_out/r4_qa_sched2_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_qa_sched2_b$b.txt; exit 1; }
  tail -5 gpurun_out/r4_qa_sched2_b$b.txt
done
