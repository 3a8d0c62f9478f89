#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_gemm -- python3 scratch/gemm_probe.py > gpurun_out/pmc_gemm.log 2>&1 || tail -5 gpurun_out/pmc_gemm.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_gemm/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    k = r['Kernel_Name'][:80]
    a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k in agg:
    if 'gemm_nt' in k and ('EpiBiasActBf16<0>' in k):
        print(k)
        for c, (n, v) in agg[k].items():
            print(f"   {c:28s} n={n:4d} avg={v/n:14.0f}")
PY
