#!/bin/bash
# SQ counters of the prompt / decode attention kernels in one short pass of the bench workload
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_attn -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --decode-tokens 8 > gpurun_out/pmc_attn.log 2>&1 || tail -5 gpurun_out/pmc_attn.log
python3 - <<'PY'
import csv, glob, collections, os
f = sorted(glob.glob('gpurun_out/pmc_attn/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r['Kernel_Name'][:60]
    a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    d = dur[k]; d[0] += 1; d[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for k in agg:
    if 'attention' in k or 'gemm_nt_glds_kernel<qasr::ADense, qasr::EpiBiasActBf16<0>, 1>' in k or 'AConv' in k:
        print(k, f"avg duration {dur[k][1]/dur[k][0]/1e3:.1f} us")
        for c, (n, v) in agg[k].items():
            print(f"   {c:28s} n={n:5d} avg={v/n:14.0f}")
PY
