#!/bin/bash
# the whole GPU suite with per-test durations, then the driver's bench command
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r4_suite.log 2>&1
rc=$?
tail -60 gpurun_out/r4_suite.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_v1.json 2> gpurun_out/r4_bench_v1.err || { tail -20 gpurun_out/r4_bench_v1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_v1.json"))
print({k: d[k] for k in ("value", "ms_per_step", "serial_inclusive_value", "resident_value", "stage_ms")})
r = d["roofline"]; print({k: r[k] for k in r if k not in ("attention", "family", "other", "how", "traffic_source", "kernel")})
print(d.get("batches")); print({k: v.get("value") if isinstance(v, dict) else v for k, v in d.get("passes_in_flight", {}).items()}); print(d.get("mlx_4bit", {}).get("value"))
print(d.get("cpu_baseline", {}).get("value"))
PY
