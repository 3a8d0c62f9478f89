#!/bin/bash
# A/B of builds inside one box (scratch/lib_alt/<name>/libqasr.so), interleaved rounds: bash scratch/lib_ab2.sh <batch> [rounds]
B=${1:-32}; R=${2:-2}
cp qwen3-asr-swift_amd/lib/libqasr.so /tmp/main.so
for round in $(seq 1 $R); do
for v in $(ls scratch/lib_alt); do
  cp scratch/lib_alt/$v/libqasr.so qwen3-asr-swift_amd/lib/libqasr.so
  echo "== $v (round $round)"
  python scratch/knob_ab.py --batch $B --rounds 3 base= 2>&1 | tail -1
done
done
cp /tmp/main.so qwen3-asr-swift_amd/lib/libqasr.so
