set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 600 python -u -m pytest tests/test_gpu_plan.py -x -q --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/r2b/plan.log | tail -15 || exit 1
timeout -k 10 600 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench.log 2>&1 || { tail -30 gpurun_out/r2b/bench.log; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2b/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], json.dumps(d['variants'])[:1500])
PY
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --eager > gpurun_out/r2b/bench_eager.log 2>&1 || { tail -30 gpurun_out/r2b/bench_eager.log; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r2b/bench_eager.log').read().strip().splitlines()[-1]); print('eager', d['value'])"
