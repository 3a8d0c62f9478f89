#!/bin/bash
# Round-4 profile of the bench workload on the GPU box: per-kernel time (rocprofv3 --kernel-trace --stats) and HBM traffic (two separate
# --pmc passes: FETCH_SIZE, WRITE_SIZE), summarised into profiles/<tag>_*.  usage: bash scratch/r03_profile.sh <tag>
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r04_v1}
rm -rf gpurun_out/prof_$TAG gpurun_out/pmcf_$TAG gpurun_out/pmcw_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err || tail -5 gpurun_out/prof_$TAG.err
cp $(ls -t $(find gpurun_out/prof_$TAG -name '*kernel_stats.csv') | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_$TAG -- python3 bench.py --steps 1 --warmup 0 --decode-tokens 16 --no-cpu-baseline --no-extras > gpurun_out/pmcf_$TAG.log 2>&1 || tail -5 gpurun_out/pmcf_$TAG.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_$TAG -- python3 bench.py --steps 1 --warmup 0 --decode-tokens 16 --no-cpu-baseline --no-extras > gpurun_out/pmcw_$TAG.log 2>&1 || tail -5 gpurun_out/pmcw_$TAG.log
python3 scratch/pmc_summarize.py gpurun_out/pmcf_$TAG gpurun_out/pmcw_$TAG $TAG
cp profiles/${TAG}_pmc_per_kernel.csv profiles/${TAG}_pmc_traffic.json gpurun_out/
find gpurun_out/prof_$TAG gpurun_out/pmcf_$TAG gpurun_out/pmcw_$TAG -name '*.csv' -size +2M -delete
head -12 gpurun_out/${TAG}_bench_kernel_stats.csv
