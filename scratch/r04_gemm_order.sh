#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scratch/knob_ab.py --batch 32 --rounds 7 --tokens 16 tm1= tm4=gemm_tm:4 tm8=gemm_tm:8 tm16=gemm_tm:16 > gpurun_out/r4_gemm_order.txt 2>&1 || { tail -5 gpurun_out/r4_gemm_order.txt; exit 1; }
tail -6 gpurun_out/r4_gemm_order.txt
