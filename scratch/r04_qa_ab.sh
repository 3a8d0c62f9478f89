#!/bin/bash
# round 4: dec_qa second form (stamps + A/B) and the non-temporal weight-load A/B (template parameters)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/r4_qa_test.log
rc=$?
cat gpurun_out/r4_qa_test.log
if [ $rc -ne 0 ]; then echo "parity failed: no A/B"; exit 1; fi
timeout -k 10 300 python scratch/chain_stamps.py 32 0 > gpurun_out/r4_qa_stamps_v2.txt 2>&1 || { tail -5 gpurun_out/r4_qa_stamps_v2.txt; exit 1; }
grep "chain" gpurun_out/r4_qa_stamps_v2.txt
for b in 32 8 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 base= qa=qa:1 nt=gemv_nt:1 lmnt=lmh_nt:1 ntboth=gemv_nt:1,lmh_nt:1 sb0=gemv_splitb:0 sb0nt=gemv_splitb:0,gemv_nt:1 > gpurun_out/r4_nt_ab_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_nt_ab_b$b.txt; exit 1; }
  tail -8 gpurun_out/r4_nt_ab_b$b.txt
done
