#!/bin/bash
# PMC passes over the split-bf16 kernel (time-only probe, 128->128, 6 products)
set -o pipefail
mkdir -p gpurun_out
export VX_PROBE_TIME_ONLY=1 VX_PROBE_MODES=6
pmc() {
  name=$1; shift
  rm -rf gpurun_out/r2h_pmc_${name}
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r2h_pmc_${name} -- \
    python3 tools/split_probe.py 8 > gpurun_out/r2h_pmc_${name}.log 2>&1 || { tail -5 gpurun_out/r2h_pmc_${name}.log; exit 1; }
  echo "pmc pass ${name} done"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES && \
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES && \
pmc tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE && \
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum && \
python3 - <<'PY'
import csv, glob, collections
for name in ("sq1", "sq2", "tcp", "tcc"):
    for f in glob.glob(f"gpurun_out/r2h_pmc_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "spconv_w8s" not in k: continue
            acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            print(name, {c: f"{v:.3e}" for c, v in d.items()})
PY
