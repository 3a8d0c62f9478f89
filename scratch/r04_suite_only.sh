#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r4_suite.log 2>&1
rc=$?
tail -22 gpurun_out/r4_suite.log
exit $rc
