#!/bin/bash
# A/B of several builds inside one box (scratch/lib_alt/<name>/libqasr.so), two rounds (drift check)
# defaults
cp qwen3-asr-swift_amd/lib/libqasr.so /tmp/main.so
for round in 1 2; do
for v in $(ls scratch/lib_alt); do
  cp scratch/lib_alt/$v/libqasr.so qwen3-asr-swift_amd/lib/libqasr.so
  echo "== $v (round $round)"
  python scratch/perf_full.py 32 2>&1 | grep -E "iter 2|probe layer"
done
done
cp /tmp/main.so qwen3-asr-swift_amd/lib/libqasr.so
