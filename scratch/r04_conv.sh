#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_gemm.py tests/test_gpu_omnilingual.py -x -q 2>&1 | tail -8 > gpurun_out/r4_conv_test.log
rc=$?
cat gpurun_out/r4_conv_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scratch/knob_ab.py --batch 32 --rounds 7 --tokens 16 kt=conv_ktile:1 chunk=conv_ktile:0 > gpurun_out/r4_conv_ab.txt 2>&1 || { tail -5 gpurun_out/r4_conv_ab.txt; exit 1; }
tail -4 gpurun_out/r4_conv_ab.txt
