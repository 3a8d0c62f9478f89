#!/bin/bash
# the whole GPU suite with per-test durations, then the driver's bench command, then the round's profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r4_suite.log 2>&1
rc=$?
tail -32 gpurun_out/r4_suite.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_v2.json 2> gpurun_out/r4_bench_v2.err || { tail -20 gpurun_out/r4_bench_v2.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_v2.json"))
print({k: d[k] for k in ("value", "ms_per_step", "serial_inclusive_value", "resident_value", "stage_ms")})
r = d["roofline"]; print({k: r[k] for k in r if k not in ("attention", "family", "other", "how", "traffic_source", "kernel")})
print(d.get("batches")); print({k: v.get("value") if isinstance(v, dict) else v for k, v in d.get("passes_in_flight", {}).items()}); print(d.get("mlx_4bit", {}).get("value"))
print(d.get("cpu_baseline", {}).get("value"), {k: v.get("value") for k, v in d.get("omnilingual", {}).items()})
PY
bash scratch/r04_profile.sh r04_v2 > gpurun_out/r4_profile_v2.log 2>&1 || { tail -20 gpurun_out/r4_profile_v2.log; exit 1; }
tail -12 gpurun_out/r4_profile_v2.log | cut -c1-220
