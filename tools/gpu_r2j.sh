#!/bin/bash
# effective clock of the split kernel and of its timing-only ablations (GRBM_GUI_ACTIVE / 8 / duration)
set -o pipefail
mkdir -p gpurun_out
export VX_PROBE_TIME_ONLY=1 VX_PROBE_MODES=6
for lib in libvx libvx_abl3 libvx_abl6; do
  rm -rf gpurun_out/r2j_$lib
  VX_LIB=$PWD/vista_amd/$lib.so timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/r2j_$lib -- \
    python3 tools/split_probe.py 8 > gpurun_out/r2j_$lib.log 2>&1 || { tail -5 gpurun_out/r2j_$lib.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for lib in ("libvx", "libvx_abl3", "libvx_abl6"):
    f = glob.glob(f"gpurun_out/r2j_{lib}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "spconv_w8s" not in r["Kernel_Name"]: continue
        d = per[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"]); d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, d in list(per.items())[-3:]:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        print(lib, f"dur {d['dur']/1e6:.3f} ms  cycles {cyc/1e6:.2f} M  clock {cyc/d['dur']:.2f} GHz  mfma busy {d['SQ_VALU_MFMA_BUSY_CYCLES']/(d['SQ_BUSY_CU_CYCLES']*4+1):.2f}")
PY
