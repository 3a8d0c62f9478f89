#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for b in 32 1; do
timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 n1= n0=qa_naps:0 n2=qa_naps:2 n4=qa_naps:4 n8=qa_naps:8 > gpurun_out/r4_naps_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_naps_b$b.txt; exit 1; }
tail -6 gpurun_out/r4_naps_b$b.txt
done
