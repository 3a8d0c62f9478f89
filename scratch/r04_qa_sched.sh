#!/bin/bash
# round 4: request schedules of the fused q|k|v + attention launch, in-situ A/B at 32 / 8 / 1 clips
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/r4_qa_test.log
rc=$?
cat gpurun_out/r4_qa_test.log
if [ $rc -ne 0 ]; then echo "parity failed: no A/B"; exit 1; fi
for b in 32 8 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 e1g0= e0g0=qa_early:0 e2g0=qa_early:2 e1g1=qa_gate:1 e0g1=qa_early:0,qa_gate:1 e2g1=qa_early:2,qa_gate:1 off=qa:0 > gpurun_out/r4_qa_sched_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_qa_sched_b$b.txt; exit 1; }
  tail -8 gpurun_out/r4_qa_sched_b$b.txt
done
