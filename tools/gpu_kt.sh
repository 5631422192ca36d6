#!/bin/bash
# kernel traces of the default inference step and the training step
set -o pipefail
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p gpurun_out
kt() {
  local name=$1; shift
  rm -rf gpurun_out/kt_${name}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_${name} -o p -- \
    python3 "$@" > gpurun_out/kt_${name}.log 2>&1 || { tail -5 gpurun_out/kt_${name}.log; exit 1; }
}
kt f32_b8 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants && \
kt train_b4 bench.py --mode train --batch 4 --steps 5 --warmup 2 && \
python3 - <<'PY'
import csv, glob
for name, steps in (("f32_b8", 25), ("train_b4", 7)):
    src = glob.glob(f"gpurun_out/kt_{name}/**/p_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {name}: {tot/1e6/steps:.2f} ms/step of kernels over {steps} steps")
    for r in rows[:22]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        print(f"{float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step {int(r['Calls'])/steps:6.1f} calls/step  {n}")
PY
tail -1 gpurun_out/kt_f32_b8.log | cut -c1-300; tail -1 gpurun_out/kt_train_b4.log | cut -c1-200
