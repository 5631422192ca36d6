set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2c/tests.log 2>&1 || { tail -40 gpurun_out/r2c/tests.log; exit 1; }
tail -3 gpurun_out/r2c/tests.log
for d in graph queued; do
timeout -k 10 600 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --dispatch $d > gpurun_out/r2c/bench_$d.log 2>&1 || { tail -30 gpurun_out/r2c/bench_$d.log; exit 1; }
python - $d <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r2c/bench_{sys.argv[1]}.log').read().strip().splitlines()[-1])
v=d['variants']
print(sys.argv[1], {k:round(d[k],2) for k in ('value','ms_per_step')}, round(d['roofline']['frac'],3), 'b1', round(v['batch1_frames_per_s'],1), 'bf16', round(v['bf16_batch8']['frames_per_s'],1), round(v['bf16_batch1']['frames_per_s'],1), 'stress', round(v['stress_batch1']['frames_per_s'],1), 'train', round(v['train_b4_ms_per_step'],1))
PY
done
timeout -k 10 300 python -u bench.py --mode train --steps 6 --warmup 2 --flat-allreduce > gpurun_out/r2c/train_flat.log 2>&1 || { tail -30 gpurun_out/r2c/train_flat.log; exit 1; }
tail -c 400 gpurun_out/r2c/train_flat.log
