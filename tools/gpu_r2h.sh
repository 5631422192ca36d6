#!/bin/bash
# PMC passes over the split-bf16 kernel (time-only probe, 128->128)
set -o pipefail
mkdir -p gpurun_out
export VX_PROBE_TIME_ONLY=1
pmc() {
  name=$1; shift
  rm -rf gpurun_out/r2h_pmc_${name}
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r2h_pmc_${name} -- \
    python3 tools/split_probe.py 8 > gpurun_out/r2h_pmc_${name}.log 2>&1 || { tail -5 gpurun_out/r2h_pmc_${name}.log; exit 1; }
  echo "pmc pass ${name} done"
}
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES && \
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES && \
python3 - <<'PY'
import csv, glob, collections
for name in ("lds", "mfma"):
    for f in glob.glob(f"gpurun_out/r2h_pmc_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "spconv_w8" not in k: continue
            k = k[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            print(name, k, {c: f"{v:.3e}" for c, v in d.items()})
PY
