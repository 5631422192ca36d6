#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_f32_products.py tests/test_gpu_parity.py tests/test_gpu_fused_bn.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 180 --timeout-method thread -k "grad or train or f32_products or fuse" > gpurun_out/r2k_tests.log 2>&1 || { tail -30 gpurun_out/r2k_tests.log; exit 1; }
tail -3 gpurun_out/r2k_tests.log
timeout -k 10 300 python bench.py --mode train --batch 4 --steps 10 --warmup 3 > gpurun_out/r2k_train.json 2> gpurun_out/r2k_train.err || { tail -20 gpurun_out/r2k_train.err; exit 1; }
cut -c1-260 gpurun_out/r2k_train.json
