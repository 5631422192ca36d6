"""Turn rocprofv3 outputs merged under gpurun_out/ into the committed summaries under profiles/.
usage: python tools/summarize_profiles.py <tag>   (expects gpurun_out/<tag>_{f32,bf16,train,full}/p_kernel_*.csv
and PMC passes gpurun_out/<q>{A,B,C,D}/vm/*_counter_collection.csv, q given as second argument)"""
import collections, csv, glob, json, re, statistics as st, sys

G = "gpurun_out"
tag, q = sys.argv[1], sys.argv[2]


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n) if len(n) > 100 else n
    return n.replace(",", ";")[:150]


def kstats(src, dst, cmd, steps):
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd}  ({steps} steps incl. warm-up, MI355X); "
                f"total GPU time {tot / steps / 1e6:.3f} ms/step\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:45]:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    print(dst, f"{tot / steps / 1e6:.2f} ms/step")


def steady(src, dst, cmd, warm, steps):
    rows = list(csv.DictReader(open(src)))
    names = [r["Kernel_Name"] for r in rows]
    s0 = [int(r["Start_Timestamp"]) for r in rows]
    e0 = [int(r["End_Timestamp"]) for r in rows]
    first = [i for i, n in enumerate(names) if "vox_insert" in n]
    i0 = first[warm]
    agg = collections.defaultdict(list)
    for i in range(i0, len(rows)):
        agg[short(names[i])].append(e0[i] - s0[i])
    tot = sum(sum(v) for v in agg.values())
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace -- python3 bench.py {cmd} ; aggregated from the kernel trace over the {steps} TIMED "
                f"steps only (the warm-up steps hold MIOpen's solver search, incl. its naive reference convolutions); "
                f"total GPU time {tot / steps / 1e6:.3f} ms/step, MI355X\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
            f.write(f"{k},{len(v)},{sum(v)},{sum(v) // len(v)},{100 * sum(v) / tot:.2f},{min(v)},{max(v)}\n")
    print(dst, f"{tot / steps / 1e6:.2f} ms/step (timed steps)")


kstats(f"{G}/{tag}_f32/p_kernel_stats.csv", "profiles/r01_bench_b8_fp32_kernel_stats.csv",
       "--steps 10 --warmup 3 --no-cpu-baseline --no-variants", 13)
kstats(f"{G}/{tag}_bf16/p_kernel_stats.csv", "profiles/r01_bench_b8_bf16_kernel_stats.csv",
       "--dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants", 13)
kstats(f"{G}/{tag}_train/p_kernel_stats.csv", "profiles/r01_train_b4_fp32_kernel_stats.csv",
       "--mode train --batch 4 --steps 5 --warmup 2  (fused libvx BatchNorm+add+ReLU kernels, the default)", 7)
steady(f"{G}/{tag}_full/p_kernel_trace.csv", "profiles/r01_fullnet_vista_b4_fp32_kernel_stats.csv",
       "--geometry VISTA --backbone MultiViewBackbone --full-net --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       3, 10)


def load(d):
    f = glob.glob(f"{G}/{d}/vm/*counter_collection.csv")[0]
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


specs = {"f32": ((q + "A", q + "B"), {"dominant_kernel": ("spconv_w8_split_tail<float, 128, 128", 12692838, 681878, 128),
                                      "c64_kernel": ("spconv_cb16<64, 64", 24411879, 1687797, 64),
                                      "c32_kernel": ("spconv_cb<float, 32, 32", 23826569, 2147993, 32)}),
         "bf16": ((q + "C", q + "D"), {"dominant_kernel": ("spconv_w8_split_tailIDF16bLi128ELi128E", 12692838, 681878, 128),
                                       "c64_kernel": ("spconv_cbIDF16bLi64ELi64E", 24411879, 1687797, 64)})}
res = {"note": "rocprofv3 --pmc passes (separate runs: FETCH_SIZE+SQ+GRBM, then WRITE_SIZE+TCC) over tools/conv_probe.py 8 "
               "[f32|bf16] on the rulebooks of 8 FHD frames; medians over the launches of each kernel. dominant_kernel = SubMConv3d "
               "128->128 k=3 (row-stationary spconv_w8 body launched through spconv_w8_split_tail: 2,664 row tiles = 10 full rounds of 256 + 104 tail tiles split over output-column halves; stage 3: 681,878 rows, 12,692,838 pairs); c64_kernel / c32_kernel = "
               "SubMConv3d 64->64 / 32->32 on the per-offset compaction kernels (fp32 64->64: spconv_cb16, 16-pair tasks + producer wave "
               "+ deferred accumulate; bf16 64->64 and 32->32: spconv_cb; stage 2: 1,687,797 rows, 24,411,879 "
               "pairs; stage 1: 2,147,993 rows, 23,826,569 pairs). FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is "
               "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane loads); GRBM_GUI_ACTIVE is "
               "summed over the 8 XCDs (kernel_cycles = sum / 8); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles x 256 CUs "
               "x 4 SIMDs).",
       "dominant_kernel": {}, "c64_kernel": {}, "c32_kernel": {}}
for dt, ((a, b), keys) in specs.items():
    A, B = load(a), load(b)
    s = 4 if dt == "f32" else 2
    for slot, (key, pairs, n_out, c) in keys.items():
        if med(A, key, "FETCH_SIZE") is None:
            print("missing", key, sorted({k[:50] for (_, k) in A if "spconv" in k}))
            continue
        fetch = med(A, key, "FETCH_SIZE") * 1024
        write = med(B, key, "WRITE_SIZE") * 1024
        cyc = med(A, key, "GRBM_GUI_ACTIVE") / 8
        mf = med(A, key, "SQ_VALU_MFMA_BUSY_CYCLES")
        hit, miss = med(B, key, "TCC_HIT_sum"), med(B, key, "TCC_MISS_sum")
        dur = med(A, key, "_dur")
        flops = 2.0 * pairs * c * c
        gbytes = pairs * c * s + n_out * c * s + 27 * c * c * s + pairs * 8
        res[slot][dt] = {"kernel": key, "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_x2_gfx950": 2 * fetch,
                         "WRITE_SIZE_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write, "kernel_cycles": cyc,
                         "mfma_busy_simd_cycles": mf, "mfma_util": mf / (cyc * 1024), "l2_hit_rate": hit / (hit + miss),
                         "lds_bank_conflict_cycles": med(A, key, "SQ_LDS_BANK_CONFLICT"),
                         "sq_wave_cycles": med(A, key, "SQ_WAVE_CYCLES"), "sq_wait_any": med(A, key, "SQ_WAIT_ANY"),
                         "sq_wait_inst_any": med(A, key, "SQ_WAIT_INST_ANY"),
                         "sq_active_inst_any": med(A, key, "SQ_ACTIVE_INST_ANY"),
                         "median_duration_ns_under_pmc": dur, "algorithmic_flops": flops, "algorithmic_gather_bytes": gbytes,
                         "algorithmic_TFLOPs_under_pmc": flops / dur / 1e3, "algorithmic_GBps_under_pmc": gbytes / dur}
        v = res[slot][dt]
        print(slot, dt, f"hbm {v['hbm_bytes_per_launch'] / 1e9:.2f} GB  mfma_util {v['mfma_util']:.3f}  l2 {v['l2_hit_rate']:.3f}  "
              f"lds_conflict {v['lds_bank_conflict_cycles']:.0f}  {v['algorithmic_TFLOPs_under_pmc']:.1f} TF  "
              f"{v['algorithmic_GBps_under_pmc']:.0f} GB/s")
json.dump(res, open("profiles/r01_pmc_summary.json", "w"), indent=1)
