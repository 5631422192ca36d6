#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_plan.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r2l_tests.log 2>&1 || { tail -30 gpurun_out/r2l_tests.log; exit 1; }
tail -2 gpurun_out/r2l_tests.log
for ov in 1 0; do
  VX_GEOM_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2l_b8_$ov.json 2> gpurun_out/r2l_b8_$ov.err || { tail -20 gpurun_out/r2l_b8_$ov.err; exit 1; }
  python - <<PY
import json
r = json.loads(open("gpurun_out/r2l_b8_$ov.json").read().strip().splitlines()[-1])
print("overlap $ov: b8", round(r["value"], 1), "ms", round(r["ms_per_step"], 2), "b1", round(r["variants"]["batch1_frames_per_s"], 1), "stress", round(r["variants"]["stress_batch1"]["frames_per_s"], 1), "bf16", round(r["variants"]["bf16_batch8"]["frames_per_s"], 1), round(r["variants"]["bf16_batch1"]["frames_per_s"], 1))
PY
done
