#!/bin/bash
# rocprofv3 kernel stats of the bench under an environment setting: prof_env.sh TAG VAR=VAL ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err || tail -5 gpurun_out/prof_$TAG.err
python3 - "$TAG" <<'PY'
import csv, glob, os, sys, json
f = sorted(glob.glob(f'gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{r['Name'][:64]:64s} avg {float(r['AverageNs'])/1e3:8.2f} us")
d = json.load(open(f'gpurun_out/prof_{sys.argv[1]}.json')); print(sys.argv[1], d['value'], d['stage_ms'])
PY
