#!/bin/bash
# round 4: the persistent layer launch (dec_chain.hip) against the five-launch layer -- parity first, then in-situ A/B at 32 / 8 / 1 clips
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/r4_chain_test.log
rc=$?
cat gpurun_out/r4_chain_test.log
if [ $rc -ne 0 ]; then echo "chain parity failed: no A/B"; exit 1; fi
for b in 32 8 1; do
  timeout -k 10 300 python scratch/knob_ab.py --batch $b --rounds 5 base= c1=chain:1 c2=chain:2 c3=chain:3 c3nt=chain:3,chain_nt:1 > gpurun_out/r4_chain_ab_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_chain_ab_b$b.txt; exit 1; }
  tail -7 gpurun_out/r4_chain_ab_b$b.txt
done
