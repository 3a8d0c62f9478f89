#!/bin/bash
# kernel-time stats only (no PMC passes) of the bench workload: bash scratch/r04_prof_quick.sh <tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-q}
rm -rf gpurun_out/profq_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profq_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profq_$TAG.json 2> gpurun_out/profq_$TAG.err || tail -5 gpurun_out/profq_$TAG.err
cp $(ls -t $(find gpurun_out/profq_$TAG -name '*kernel_stats.csv') | head -1) gpurun_out/${TAG}_quick_kernel_stats.csv
find gpurun_out/profq_$TAG -name '*.csv' -size +2M -delete
grep -E "greedy_finalize|lm_head_kernel|decode_qa_kernel|decode_gemv2" gpurun_out/${TAG}_quick_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
