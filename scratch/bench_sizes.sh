#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in 1 8; do
  python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err || tail -5 gpurun_out/bench_b$b.err
  python -c "import json; d=json.load(open('gpurun_out/bench_b$b.json')); print('B=$b', d['value'], d['ms_per_step'], d['stage_ms'])"
done
