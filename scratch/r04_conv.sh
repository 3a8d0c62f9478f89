#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -8 > gpurun_out/r4_conv_test.log
rc=$?
cat gpurun_out/r4_conv_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python scratch/knob_ab.py --batch 32 --rounds 5 --tokens 16 --allow-token-drift kb=conv_korder:1 tap=conv_korder:0 > gpurun_out/r4_conv_ab.txt 2>&1 || { tail -5 gpurun_out/r4_conv_ab.txt; exit 1; }
tail -4 gpurun_out/r4_conv_ab.txt
export TMPDIR=/tmp
for o in 1 0; do
  rm -rf gpurun_out/pmc_conv_$o
  QASR_CONV_KORDER=$o rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_conv_$o -- python3 bench.py --steps 1 --warmup 0 --decode-tokens 2 --no-cpu-baseline --no-extras > gpurun_out/pmc_conv_$o.log 2>&1 || tail -5 gpurun_out/pmc_conv_$o.log
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/pmc_conv_$o/**/*counter_collection.csv", recursive=True))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "AConv3x3s2" in r["Kernel_Name"]:
        acc.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("conv_korder=$o", k, "launches", len(v), "fetched GB per launch (2 x FETCH_SIZE KB):", [round(2 * x * 1024 / 1e9, 2) for x in v])
PY
  find gpurun_out/pmc_conv_$o -name '*.csv' -size +2M -delete
done
