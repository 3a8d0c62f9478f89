#!/bin/bash
# SQ counters of the two flash-attention kernels at their production shapes: prefill_attention2_kernel (head_dim 128,
# causal GQA, 0.13 of the MFMA peak) against mha64_attention_kernel (head_dim 64, 0.28) -- two passes of 4 counters each.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_pa_$tag
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_pa_$tag -- python3 bench.py --steps 1 --warmup 0 --decode-tokens 4 --no-cpu-baseline --no-extras > gpurun_out/pmc_pa_$tag.log 2>&1 || tail -3 gpurun_out/pmc_pa_$tag.log
done
python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in glob.glob('gpurun_out/pmc_pa_*/'):
    fs = sorted(glob.glob(d + '**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name']
        if 'prefill_attention2' in k: k = 'prefill_attention2_kernel<128>'
        elif 'mha64_attention' in k: k = 'mha64_attention_kernel'
        else: continue
        a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
with open('gpurun_out/r03_pmc_attention_kernels.txt', 'w') as f:
    for k in agg:
        f.write(k + '\n')
        for c, (n, v) in sorted(agg[k].items()):
            f.write(f"   {c:28s} launches={n:5d} avg_per_launch={v/n:16.0f}\n")
print(open('gpurun_out/r03_pmc_attention_kernels.txt').read())
PY
find gpurun_out/pmc_pa_* -name '*.csv' -size +1M -delete
