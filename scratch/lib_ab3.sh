#!/bin/bash
# checksum + timing of builds inside one box (scratch/lib_alt/<name>/libqasr.so)
cp qwen3-asr-swift_amd/lib/libqasr.so /tmp/main.so
for v in $(ls scratch/lib_alt); do
  cp scratch/lib_alt/$v/libqasr.so qwen3-asr-swift_amd/lib/libqasr.so
  echo "== $v"
  python scratch/enc_checksum.py 2>&1 | grep -E "^encoder|^tokens"
  python scratch/knob_ab.py --batch 32 --tokens 8 --rounds 5 base= 2>&1 | tail -1
done
cp /tmp/main.so qwen3-asr-swift_amd/lib/libqasr.so
