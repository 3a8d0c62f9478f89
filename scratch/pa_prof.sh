#!/bin/bash
# per-kernel time of the prompt pass for the two V operand forms of the prompt attention (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in 0 1; do
  export QASR_PA_VFRAG=$v
  rm -rf gpurun_out/paprof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/paprof_$v -- python3 bench.py --steps 3 --warmup 1 --decode-tokens 8 --no-cpu-baseline --no-extras > gpurun_out/paprof_$v.json 2> gpurun_out/paprof_$v.err || tail -5 gpurun_out/paprof_$v.err
  f=$(ls -t $(find gpurun_out/paprof_$v -name '*kernel_stats.csv') | head -1)
  echo "== pa_vfrag=$v"; grep -E "prefill_attention|v_transpose|qk_norm_rope" $f | cut -c1-60,200-400
  cp $f gpurun_out/paprof_${v}_kernel_stats.csv
  find gpurun_out/paprof_$v -name '*.csv' -size +2M -delete
done
