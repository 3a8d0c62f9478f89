#!/bin/bash
# per-kernel time of the bench workload at another batch size: bash scratch/r04_prof_b.sh <batch>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B=${1:-1}
rm -rf gpurun_out/prof_b$B
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -- python3 bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_b$B.json 2> gpurun_out/prof_b$B.err || tail -5 gpurun_out/prof_b$B.err
cp $(ls -t $(find gpurun_out/prof_b$B -name '*kernel_stats.csv') | head -1) gpurun_out/r04_b${B}_kernel_stats.csv
find gpurun_out/prof_b$B -name '*.csv' -size +2M -delete
head -12 gpurun_out/r04_b${B}_kernel_stats.csv | cut -c1-220
