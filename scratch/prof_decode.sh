#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python3 scratch/perf_full.py ${1:-32} ${2:-32} ${3:-30} > gpurun_out/prof_dec.log 2>&1
tail -6 gpurun_out/prof_dec.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_dec/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
