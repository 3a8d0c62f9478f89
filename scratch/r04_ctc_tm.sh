#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python scratch/bench_ctc.py --variant 300M --steps 5 --ab gemm_tm=1,8,1,8,4,16 > gpurun_out/r4_ctc_tm_300m.txt 2>&1 || { tail -5 gpurun_out/r4_ctc_tm_300m.txt; exit 1; }
grep -v amdgpu gpurun_out/r4_ctc_tm_300m.txt | tail -8 | cut -c1-400
