#!/bin/bash
# Round-2 rocprofv3 runs (through gpurun from the repo root): kernel traces of the bench configurations and HBM
# traffic counters of the hash / rulebook / dense kernels. Kernel traces and counter passes are separate runs; each
# counter pass stays within one block's limits and the program follows `--` directly.
set -e
tag=r02
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
kt hash_fhd8 tools/hash_probe.py FHD 8 3
kt hash_stress1 tools/hash_probe.py STRESS 1 3
pmc() {
  local name=$1 geom=$2 batch=$3; shift 3
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -- \
    python3 tools/hash_probe.py $geom $batch 3 > gpurun_out/${tag}_pmc_${name}.log 2>&1
  echo "pmc pass ${name} done"
}
pmc fhd8_fetch FHD 8 FETCH_SIZE
pmc fhd8_write FHD 8 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc stress1_fetch STRESS 1 FETCH_SIZE
pmc stress1_write STRESS 1 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls gpurun_out | grep ${tag}_ | head -40
