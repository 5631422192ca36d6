"""Turn the rocprofv3 outputs of tools/run_profiles_r02b.sh (merged under gpurun_out/) into the committed summaries under
profiles/: kernel-stat tables of the bench configurations and profiles/r02_pmc_summary.json (counters of the convolution
kernels in the default split-bf16 arithmetic and on the fp32 MFMA)."""
import csv
import glob
import json
import os
import re
import statistics as st

G = "gpurun_out"
# round 3 re-ran the counter passes only (tools/run_pmc_r03.sh): VX_PROF_TAG=r03 VX_PROF_PMC_ONLY=1 VX_PROF_OUT=profiles/r03_pmc_summary.json
T = os.environ.get("VX_PROF_TAG", "r02b")
PMC_ONLY = bool(os.environ.get("VX_PROF_PMC_ONLY"))
OUT = os.environ.get("VX_PROF_OUT", "profiles/r02_pmc_summary.json")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace(",", ";")[:110]


def kstats(tag, dst, cmd, note):
    src = glob.glob(f"{G}/{T}_{tag}/**/p_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}  (MI355X; {note}); total GPU time {tot / 1e6:.1f} ms\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:45]:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    print(dst, f"{tot / 1e6:.1f} ms total")


ARITH = "default fp32 arithmetic = 6 split-bf16 plane products"
if PMC_ONLY:
    kstats = lambda *a, **k: None
kstats("f32_b8", "profiles/r02_bench_b8_fp32_kernel_stats.csv", "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       f"{ARITH}; 3 warm-up + 10 timed steps as one captured HIP graph each, a probe pass and 1 + 10 instrumented eager steps: 25 steps of kernels")
kstats("f32_b1", "profiles/r02_bench_b1_fp32_kernel_stats.csv", "bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants",
       f"{ARITH}; 3 + 20 graph steps, a probe pass and 1 + 20 instrumented eager steps: 45 steps of kernels")
kstats("bf16_b8", "profiles/r02_bench_b8_bf16_kernel_stats.csv", "bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       "25 steps of kernels as for fp32")
kstats("train_b4", "profiles/r02_train_b4_fp32_kernel_stats.csv", "bench.py --mode train --batch 4 --steps 5 --warmup 2",
       f"{ARITH} in fwd, dgrad and wgrad; 7 training steps, fused BatchNorm kernels, gradients in one flat buffer")


def load(d):
    f = glob.glob(f"{G}/{T}_pmc_{d}/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        c = per.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})
        c[r["Counter_Name"]] = float(r["Counter_Value"])
        c["_grid"] = int(r["Grid_Size"])
        c["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


def med(per, key, ctr):
    """median over the launches of `key` with the largest grid (the SubM layer; smaller grids are other layers)"""
    sel = [c for (d, k), c in per.items() if key in k and ctr in c]
    if not sel:
        return None
    g = max(c["_grid"] for c in sel)
    return st.median([c[ctr] for c in sel if c["_grid"] == g])


L3 = (12692838, 681878, 128)      # pairs, rows, channels of the stage-3 SubM layer (8 FHD frames)
L2 = (24411879, 1687797, 64)
L1 = (23826569, 2147993, 32)
specs = {"f32_split6": (("s6_A", "s6_B", "s6_C"), {"dominant_kernel": ("spconv_w8s_split_tail<128, 128, 6", *L3),
                                                    "c64_kernel": ("spconv_cb16<64, 64, 16, 6", *L2),
                                                    "c32_kernel": ("spconv_cb<float, 32, 32", *L1)}),
         "f32": (("m0_A", "m0_B", None), {"dominant_kernel": ("spconv_w8_split_tail<float, 128, 128", *L3),
                                          "c64_kernel": ("spconv_cb16<64, 64, 16, 0", *L2)})}
res = {"note": "rocprofv3 --pmc passes (separate runs: FETCH_SIZE+SQ+GRBM, WRITE_SIZE+TCC, TA/TCP/TD/LDS) over tools/conv_probe.py 8 on "
               "the rulebooks of 8 FHD frames; medians over the launches of each kernel. f32_split6 = the library default (every fp32 "
               "operand split exactly into three bf16 planes, six plane products on the bf16 matrix pipe, fp32 accumulate); f32 = the "
               "same layers on the fp32 MFMA (VX_F32_PRODUCTS=0). dominant_kernel = SubMConv3d 128->128 k=3 (stage 3: 681,878 rows, "
               "12,692,838 pairs), c64_kernel = SubMConv3d 64->64 on the 16-pair compaction kernel (stage 2: 1,687,797 rows, "
               "24,411,879 pairs), c32_kernel = SubMConv3d 32->32 on the 32-pair compaction kernel (fp32 MFMA in both modes). "
               "FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
               "requests at 64 B for 16 B/lane loads); GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel_cycles = sum / 8, "
               "clock_GHz = kernel_cycles / duration); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs); "
               "ta_busy / td_busy / tcp_pending_stall / lds_idx_active are fractions of CU-cycles (sum over CUs / SQ_BUSY_CU_CYCLES).",
       "dominant_kernel": {}, "c64_kernel": {}, "c32_kernel": {}}
for mode, ((a, b, c3), keys) in specs.items():
    A, B = load(a), load(b)
    C = load(c3) if c3 else None
    for slot, (key, pairs, n_out, c) in keys.items():
        if med(A, key, "FETCH_SIZE") is None:
            print("missing", key, sorted({k[:60] for (_, k) in A if "spconv" in k}))
            continue
        fetch = med(A, key, "FETCH_SIZE") * 1024
        write = med(B, key, "WRITE_SIZE") * 1024
        cyc = med(A, key, "GRBM_GUI_ACTIVE") / 8
        mf = med(A, key, "SQ_VALU_MFMA_BUSY_CYCLES")
        cu = med(A, key, "SQ_BUSY_CU_CYCLES")
        hit, miss = med(B, key, "TCC_HIT_sum"), med(B, key, "TCC_MISS_sum")
        dur = med(A, key, "_dur")
        flops = 2.0 * pairs * c * c
        gbytes = pairs * c * 4 + n_out * c * 4 + 27 * c * c * 4 + pairs * 8
        v = {"kernel": key, "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_x2_gfx950": 2 * fetch, "WRITE_SIZE_bytes": write,
             "hbm_bytes_per_launch": 2 * fetch + write, "kernel_cycles": cyc, "clock_GHz_under_pmc": cyc / dur,
             "mfma_busy_simd_cycles": mf, "mfma_util": mf / (cu * 4), "l2_hit_rate": hit / (hit + miss),
             "lds_bank_conflict_cycles": med(A, key, "SQ_LDS_BANK_CONFLICT"), "sq_busy_cu_cycles": cu,
             "sq_wave_cycles": med(A, key, "SQ_WAVE_CYCLES"), "sq_wait_any": med(A, key, "SQ_WAIT_ANY"),
             "sq_wait_inst_any": med(A, key, "SQ_WAIT_INST_ANY"), "sq_active_inst_any": med(A, key, "SQ_ACTIVE_INST_ANY"),
             "median_duration_ns_under_pmc": dur, "algorithmic_flops": flops, "algorithmic_gather_bytes": gbytes,
             "algorithmic_TFLOPs_under_pmc": flops / dur / 1e3, "algorithmic_GBps_under_pmc": gbytes / dur}
        if C is not None and med(C, key, "TA_TA_BUSY_sum") is not None:
            cu3 = med(C, key, "SQ_BUSY_CU_CYCLES")
            v.update({"ta_busy": med(C, key, "TA_TA_BUSY_sum") / cu3,
                      "ta_addr_stalled_by_tc": med(C, key, "TA_ADDR_STALLED_BY_TC_CYCLES_sum") / cu3,
                      "tcp_pending_stall": med(C, key, "TCP_PENDING_STALL_CYCLES_sum") / cu3,
                      "td_busy": med(C, key, "TD_TD_BUSY_sum") / cu3,
                      "lds_idx_active": med(C, key, "SQ_LDS_IDX_ACTIVE") / cu3,
                      "valu_insts_per_launch": med(C, key, "SQ_INSTS_VALU")})
        res[slot][mode] = v
        print(slot, mode, f"hbm {v['hbm_bytes_per_launch'] / 1e9:.2f} GB  mfma_util {v['mfma_util']:.3f}  l2 {v['l2_hit_rate']:.3f}  "
              f"clock {v['clock_GHz_under_pmc']:.2f} GHz  lds_conflict {v['lds_bank_conflict_cycles']:.0f}  "
              f"{v['algorithmic_TFLOPs_under_pmc']:.1f} TF  {v['algorithmic_GBps_under_pmc']:.0f} GB/s")
json.dump(res, open(OUT, "w"), indent=1)
