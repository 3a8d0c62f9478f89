#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for b in 32; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 e3=qa_early:3 e6=qa_early:6 e7=qa_early:7 > gpurun_out/r4_qa_sched3_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_qa_sched3_b$b.txt; exit 1; }
  tail -4 gpurun_out/r4_qa_sched3_b$b.txt
done
