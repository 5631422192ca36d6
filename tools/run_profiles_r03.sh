#!/bin/bash
# Round-3 rocprofv3 runs (through gpurun from the repo root): kernel traces of the bench configurations. The conv, rulebook
# and voxelizer kernels are those of round 2 (this round's kernel experiments were measured and parked),
# so the round-2 counter passes (profiles/r02_pmc_*.json) still describe them; the traces below re-measure the step.
set -e
tag=r03
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
kt() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name} -o p -- \
    python3 "$@" > gpurun_out/${tag}_${name}.log 2>&1
  echo "kernel trace ${name} done"
}
kt f32_b8 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants
kt f32_b1 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants
kt bf16_b8 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants
kt train_b4 bench.py --mode train --batch 4 --steps 5 --warmup 2
ls gpurun_out | grep ${tag}_ | head -40
