#!/bin/bash
# round 4: stamps of dec_qa / dec_chain variants, in-situ A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/r4_qa_test.log
rc=$?
cat gpurun_out/r4_qa_test.log
if [ $rc -ne 0 ]; then echo "parity failed: no A/B"; exit 1; fi
timeout -k 10 300 python scratch/chain_stamps.py 32 0 > gpurun_out/r4_qa_stamps.txt 2>&1 || { tail -5 gpurun_out/r4_qa_stamps.txt; exit 1; }
grep "chain" gpurun_out/r4_qa_stamps.txt
timeout -k 10 300 python scratch/chain_stamps.py 32 3 chain_proto:1 chain_pf:1 > gpurun_out/r4_chain_stamps_v2.txt 2>&1 || { tail -5 gpurun_out/r4_chain_stamps_v2.txt; exit 1; }
grep "chain" gpurun_out/r4_chain_stamps_v2.txt
for b in 32 1; do
  timeout -k 10 400 python scratch/knob_ab.py --batch $b --rounds 5 base= c2=chain:2 c2r=chain:2,chain_proto:1 c2p=chain:2,chain_pf:1 c2rp=chain:2,chain_proto:1,chain_pf:1 c3rp=chain:3,chain_proto:1,chain_pf:1 qa=qa:1 > gpurun_out/r4_v2_ab_b$b.txt 2>&1 || { tail -5 gpurun_out/r4_v2_ab_b$b.txt; exit 1; }
  tail -8 gpurun_out/r4_v2_ab_b$b.txt
done
