#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2m_split.log 2>&1 || { tail -30 gpurun_out/r2m_split.log; exit 1; }
grep -E "products" gpurun_out/r2m_split.log | cut -c1-140
timeout -k 10 600 python -u -m pytest tests/test_gpu_f32_products.py tests/test_gpu_plan.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r2m_tests.log 2>&1 || { tail -30 gpurun_out/r2m_tests.log; exit 1; }
tail -2 gpurun_out/r2m_tests.log
