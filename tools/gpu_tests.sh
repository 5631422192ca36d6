#!/bin/bash
# full GPU suite, one process
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread "$@" > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests.log
exit $rc
