#!/bin/bash
# Round-2 rocprofv3 runs with the split-bf16 kernels (through gpurun from the repo root): kernel traces of the bench
# configurations and counter passes over the convolution kernels (tools/conv_probe.py 8 on the rulebooks of 8 FHD
# frames) in the default arithmetic (6 plane products) and on the fp32 MFMA (VX_F32_PRODUCTS=0). Kernel traces and
# counter passes are separate runs; each pass stays within one block's limits; the program follows `--` directly.
set -o pipefail
tag=r02b
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p gpurun_out
kt() {
  local name=$1; shift
  rm -rf gpurun_out/${tag}_${name}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name} -o p -- \
    python3 "$@" > gpurun_out/${tag}_${name}.log 2>&1 || { tail -5 gpurun_out/${tag}_${name}.log; exit 1; }
  echo "kernel trace ${name} done"
}
pmc() {
  local name=$1; shift
  rm -rf gpurun_out/${tag}_pmc_${name}
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -- \
    python3 tools/conv_probe.py 8 > gpurun_out/${tag}_pmc_${name}.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_${name}.log; exit 1; }
  echo "pmc pass ${name} done"
}
SQ="SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
kt f32_b8 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants && \
kt f32_b1 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants && \
kt bf16_b8 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants && \
kt train_b4 bench.py --mode train --batch 4 --steps 5 --warmup 2 && \
pmc s6_A FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_BF16 && \
pmc s6_B WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && \
pmc s6_C TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE && \
export VX_F32_PRODUCTS=0 && \
pmc m0_A FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_F32 && \
pmc m0_B WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && \
ls gpurun_out | grep ${tag}_ | head -40
