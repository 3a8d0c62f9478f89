#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_full.py tests/test_gpu_aligner.py tests/test_gpu_decoder.py -x -q 2>&1 | tail -6 > gpurun_out/r4_fuseqk_test.log
rc=$?
cat gpurun_out/r4_fuseqk_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scratch/knob_ab.py --batch 32 --rounds 7 --tokens 16 fused= apart=pp_fuse_qk:0 > gpurun_out/r4_fuseqk_ab.txt 2>&1 || { tail -5 gpurun_out/r4_fuseqk_ab.txt; exit 1; }
tail -3 gpurun_out/r4_fuseqk_ab.txt
timeout -k 10 400 python scratch/knob_ab.py --batch 1 --rounds 7 --tokens 16 fused= apart=pp_fuse_qk:0 > gpurun_out/r4_fuseqk_ab1.txt 2>&1 || { tail -5 gpurun_out/r4_fuseqk_ab1.txt; exit 1; }
tail -3 gpurun_out/r4_fuseqk_ab1.txt
