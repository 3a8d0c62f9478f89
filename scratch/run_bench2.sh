#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the same workload
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01_v3}
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -20 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err || { tail -20 gpurun_out/prof_$TAG.err; exit 1; }
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --decode-tokens 16 > gpurun_out/pmc_fetch_$TAG.json 2> gpurun_out/pmc_fetch_$TAG.err || { tail -20 gpurun_out/pmc_fetch_$TAG.err; exit 1; }
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --decode-tokens 16 > gpurun_out/pmc_write_$TAG.json 2> gpurun_out/pmc_write_$TAG.err || { tail -20 gpurun_out/pmc_write_$TAG.err; exit 1; }
echo "pmc write done"
find gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG -name "*.csv" | head
