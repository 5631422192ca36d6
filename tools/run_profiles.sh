#!/bin/bash
# Collects the rocprofv3 runs behind profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/run_profiles.sh r3 q3'
# then, back in the container:  python tools/summarize_profiles.py r3 q3
# Kernel traces and counter passes are separate runs; each counter pass stays within one block's limits
# (8 SQ, 4 TCC: FETCH_SIZE uses 3, WRITE_SIZE 2) and the program follows `--` directly.
set -e
tag=$1; q=$2
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
kt() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name} -o p -- \
    python3 bench.py "$@" > gpurun_out/${tag}_${name}.log 2>&1
  echo "kernel trace ${name} done"
}
kt f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants
kt bf16 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants
kt train --mode train --batch 4 --steps 5 --warmup 2
kt full --geometry VISTA --backbone MultiViewBackbone --full-net --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-variants
pmc() {  # suffix, dtype, counters...
  local sfx=$1 dt=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${q}${sfx} -- \
    python3 tools/conv_probe.py 8 $dt > gpurun_out/${q}${sfx}.log 2>&1
  echo "pmc pass ${sfx} done"
}
SQ="SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
pmc A f32 FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_F32
pmc B f32 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc C bf16 FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_BF16
pmc D bf16 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls gpurun_out/${q}A/vm gpurun_out/${tag}_f32
