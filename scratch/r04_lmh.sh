#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for b in 32 1; do
timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 7 wgfast= wavefast=lmh_order:0 > gpurun_out/r4_lmh_ab_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_lmh_ab_b$b.txt; exit 1; }
tail -3 gpurun_out/r4_lmh_ab_b$b.txt
done
