#!/bin/bash
# bias-gradient skip + from_dense: tests, then the training bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_fused_bn.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2e_tests.log 2>&1 || { tail -30 gpurun_out/r2e_tests.log; exit 1; }
tail -3 gpurun_out/r2e_tests.log
timeout -k 10 300 python bench.py --mode train --batch 4 --steps 10 --warmup 3 > gpurun_out/r2e_train.json 2> gpurun_out/r2e_train.err || { tail -20 gpurun_out/r2e_train.err; exit 1; }
cat gpurun_out/r2e_train.json
