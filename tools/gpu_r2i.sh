#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/conv_probe.py 8 > gpurun_out/r2i_cb.log 2>&1 || { tail gpurun_out/r2i_cb.log; exit 1; }
grep -- "->" gpurun_out/r2i_cb.log | cut -c1-120
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_f32_products.py -m gpu -x -q --timeout 180 --timeout-method thread -k "conv or compaction or f32" > gpurun_out/r2i_tests.log 2>&1 || { tail -30 gpurun_out/r2i_tests.log; exit 1; }
tail -3 gpurun_out/r2i_tests.log
