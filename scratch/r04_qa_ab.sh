#!/bin/bash
# round 4: q|k|v + attention in one launch (dec_qa.hip), chain stamps, in-situ A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/r4_qa_test.log
rc=$?
cat gpurun_out/r4_qa_test.log
if [ $rc -ne 0 ]; then echo "parity failed: no A/B"; exit 1; fi
timeout -k 10 300 python scratch/chain_stamps.py 32 3 > gpurun_out/r4_chain_stamps.txt 2>&1 || { tail -5 gpurun_out/r4_chain_stamps.txt; exit 1; }
grep "chain" gpurun_out/r4_chain_stamps.txt
for b in 32 8 1; do
  timeout -k 10 300 python scratch/knob_ab.py --batch $b --rounds 5 base= qa=qa:1 qac2=qa:1,chain:2 > gpurun_out/r4_qa_ab_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_qa_ab_b$b.txt; exit 1; }
  tail -5 gpurun_out/r4_qa_ab_b$b.txt
done
