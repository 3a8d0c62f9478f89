#!/bin/bash
# first measured bench line + rocprofv3 kernel stats of the same command
set -e
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || { tail -20 gpurun_out/bench_r01.err; exit 1; }
cat gpurun_out/bench_r01.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || { tail -20 gpurun_out/prof_bench.err; exit 1; }
find gpurun_out/prof_r01 -name "*kernel_stats*" | head
