#!/bin/bash
# Round-3 counter passes over the convolution kernels (tools/conv_probe.py 8 on the rulebooks of 8 FHD frames), default
# arithmetic and fp32 MFMA: the same passes as tools/run_profiles_r02b.sh, re-collected on this round's library.
# Summary: VX_PROF_TAG=r03 VX_PROF_PMC_ONLY=1 VX_PROF_OUT=profiles/r03_pmc_summary.json python tools/summarize_profiles_r02b.py
set -o pipefail
tag=r03
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p gpurun_out
pmc() {
  local name=$1; shift
  rm -rf gpurun_out/${tag}_pmc_${name}
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -- \
    python3 tools/conv_probe.py 8 > gpurun_out/${tag}_pmc_${name}.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_${name}.log; exit 1; }
  echo "pmc pass ${name} done"
}
SQ="SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
pmc s6_A FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_BF16 && \
pmc s6_B WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && \
pmc s6_C TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE && \
export VX_F32_PRODUCTS=0 && \
pmc m0_A FETCH_SIZE GRBM_GUI_ACTIVE $SQ SQ_INSTS_VALU_MFMA_MOPS_F32 && \
pmc m0_B WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
