set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 900 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused_bn.py tests/test_gpu_parity.py -x -q --timeout 400 --timeout-method thread -k "vs_oracle or fuse_for_training" --durations=8 -s 2>&1 | tee gpurun_out/r2d/tests.log | tail -25
