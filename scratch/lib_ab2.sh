#!/bin/bash
# A/B of builds inside one box (scratch/lib_alt/<name>/libqasr.so), two rounds: stage times of the bench pass (16 tokens) and of Omnilingual 300M
cd "$GRAFT_REPO_ROOT"
cp qwen3-asr-swift_amd/lib/libqasr.so /tmp/main.so
for round in 1 2; do
for v in $(ls scratch/lib_alt); do
  cp scratch/lib_alt/$v/libqasr.so qwen3-asr-swift_amd/lib/libqasr.so
  echo "== $v (round $round)"
  python scratch/perf_full.py 32 16 2>&1 | grep -E "iter 2"
  python scratch/bench_ctc.py --variant 300M --batch 32 --seconds 30 2>&1 | tail -1 | cut -c1-200
done
done
cp /tmp/main.so qwen3-asr-swift_amd/lib/libqasr.so
