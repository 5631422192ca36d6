#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("achieved", "peak", "frac", "avg_launch_ms") if k in r.get("roofline", {})}, r["roofline"].get("frac_of_fp32_mfma_peak"))
for k, v in r.get("variants", {}).items():
    print(k, v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("frames_per_s", "ms_per_step")})
print("cpu", r.get("cpu_baseline"))
PY
