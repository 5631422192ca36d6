#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export VX_PROBE_TIME_ONLY=1 VX_PROBE_MODES=6
rm -rf gpurun_out/r2o
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r2o -- python3 tools/split_probe.py 8 > gpurun_out/r2o.log 2>&1 || { tail -5 gpurun_out/r2o.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r2o/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "spconv_w8s" not in r["Kernel_Name"]: continue
    d = per[r["Dispatch_Id"]]; d[r["Counter_Name"]] = float(r["Counter_Value"]); d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, d in list(per.items())[-2:]:
    print({c: f"{v:.3e}" for c, v in d.items()}, "conflict/cu", d["SQ_LDS_BANK_CONFLICT"] / d["SQ_BUSY_CU_CYCLES"], "clock", d["GRBM_GUI_ACTIVE"] / 8 / d["dur"])
PY
