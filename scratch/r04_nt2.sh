#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for b in 32 8 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 qa= qant=qa_nt:1 off=qa:0 offnt=qa:0,da_nt:1 > gpurun_out/r4_kvnt_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_kvnt_b$b.txt; exit 1; }
  tail -5 gpurun_out/r4_kvnt_b$b.txt
done
