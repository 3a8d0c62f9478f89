#!/bin/bash
# A/B of two builds of libqasr.so on one box: lib/libqasr.so (new) against lib/libqasr_old.so, stage times at 32, 8 and 1 clips, new-old-new-old
cd "$GRAFT_REPO_ROOT"
L=qwen3-asr-swift_amd/lib
cp $L/libqasr.so $L/libqasr_new.so
for rep in 1 2; do
  for v in new old; do
    cp $L/libqasr_$v.so $L/libqasr.so
    for b in 32 8 1; do
      echo "== $v rep $rep batch $b"
      timeout -k 10 200 python scratch/knob_ab.py --batch $b --rounds 3 base= 2>/dev/null | tail -1
    done
  done
done
cp $L/libqasr_new.so $L/libqasr.so
