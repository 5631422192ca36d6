"""Turn the rocprofv3 outputs of tools/run_profiles_r02.sh (merged under gpurun_out/) into the committed summaries
under profiles/: kernel-stat tables and profiles/r02_pmc_hash.json (HBM traffic of the hash / rulebook / dense kernels)."""
import collections
import csv
import glob
import json
import re

G = "gpurun_out"


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace(",", ";")[:110]


def kstats(tag, dst, cmd, note):
    src = glob.glob(f"{G}/r02_{tag}/**/p_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}  (MI355X; {note}); total GPU time {tot / 1e6:.1f} ms\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:45]:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    print(dst, f"{tot / 1e6:.1f} ms total")


kstats("f32_b8", "profiles/r02_bench_b8_fp32_kernel_stats.csv", "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       "3 warm-up + 10 timed steps as one captured HIP graph each, a probe pass and 1 + 10 instrumented eager steps: 25 steps of kernels")
kstats("f32_b1", "profiles/r02_bench_b1_fp32_kernel_stats.csv", "bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants",
       "3 + 20 graph steps, a probe pass and 1 + 20 instrumented eager steps: 45 steps of kernels")
kstats("bf16_b8", "profiles/r02_bench_b8_bf16_kernel_stats.csv", "bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       "25 steps of kernels as for fp32")
kstats("train_b4", "profiles/r02_train_b4_fp32_kernel_stats.csv", "bench.py --mode train --batch 4 --steps 5 --warmup 2",
       "7 training steps, fused BatchNorm kernels, gradients in one flat buffer")
kstats("hash_fhd8", "profiles/r02_hash_fhd8_kernel_stats.csv", "tools/hash_probe.py FHD 8 3", "3 geometry passes over 8 FHD frames")
kstats("hash_stress1", "profiles/r02_hash_stress1_kernel_stats.csv", "tools/hash_probe.py STRESS 1 3", "3 geometry passes over 1 STRESS frame")

PHASE = {"vox_init_table": "voxelize", "vox_insert": "voxelize", "vox_flag": "voxelize", "vox_assign": "voxelize",
         "vox_frame_counts": "voxelize", "vox_fill_lists": "voxelize", "vox_gather": "voxelize",
         "hash_clear": "rulebook_subm", "hash_build": "rulebook_subm", "subm_nbr": "rulebook_subm",
         "subm_nbr_ranked": "rulebook_subm", "conv_mark": "rulebook_conv", "bitmap_pack": "rulebook_conv",
         "conv_emit_coords": "rulebook_conv", "rank_interleave": "rulebook_conv", "conv_pairs_from_inputs": "rulebook_conv",
         "to_dense_kernel<unsigned int>": "dense"}
SHARED = ("fill_kernel", "scan_reduce_kernel", "scan_apply_kernel")


def counters(tag):
    """-> list of dispatches in order: (kernel, {counter: value}) merged over the fetch and write passes."""
    per = {}
    for kind in ("fetch", "write"):
        f = glob.glob(f"{G}/r02_pmc_{tag}_{kind}/**/*counter_collection.csv", recursive=True)[0]
        seq = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            seq.setdefault(int(r["Dispatch_Id"]), [short(r["Kernel_Name"]), {}])[1][r["Counter_Name"]] = float(r["Counter_Value"])
        per[kind] = [v for _, v in sorted(seq.items())]
    ours = lambda seq: [d for d in seq if d[0] in PHASE or d[0] in SHARED]  # noqa: E731
    a, b = ours(per["fetch"]), ours(per["write"])
    assert [d[0] for d in a] == [d[0] for d in b], "the two passes launched different kernel sequences"
    return [(x[0], {**x[1], **y[1]}) for x, y in zip(a, b)]


out = {"note": "rocprofv3 --pmc over tools/hash_probe.py (3 geometry passes: voxelization + mean VFE, the 4 submanifold and 4 "
               "strided rulebooks of SpMiddleResNetFHD, dense()); pass 1 FETCH_SIZE, pass 2 WRITE_SIZE + TCC_HIT_sum + "
               "TCC_MISS_sum. FETCH_SIZE / WRITE_SIZE come in KiB; hbm_bytes = 2 x FETCH_SIZE (the gfx950 correction of "
               "MI355X_MICROARCH.md for 16-byte-per-lane reads; the 4- and 8-byte accesses of these kernels are NOT "
               "calibrated, so fetch_bytes_raw is given too: the true read traffic lies between the two) + WRITE_SIZE. "
               "Shared helper kernels (fill, scan) are booked to the phase of the next kernel in the stream.",
       "per_kernel": {}, "per_frame_bytes": {}, "per_frame_detail": {}}
for tag, geom, batch in (("fhd8", "FHD", 8), ("stress1", "STRESS", 1)):
    seq = counters(tag)
    reps = 3
    phase_of = [None] * len(seq)
    nxt = None
    for i in range(len(seq) - 1, -1, -1):
        if seq[i][0] in PHASE:
            nxt = PHASE[seq[i][0]]
        phase_of[i] = nxt
    kern = collections.defaultdict(lambda: collections.defaultdict(float))
    phase = collections.defaultdict(lambda: collections.defaultdict(float))
    for (name, c), ph in zip(seq, phase_of):
        for k, v in c.items():
            kern[name][k] += v
            phase[ph][k] += v
        kern[name]["launches"] += 1
    key = f"{geom}_b{batch}"
    out["per_kernel"][key] = {
        n: {"launches_per_pass": c["launches"] / reps, "fetch_bytes_raw": 1024 * c["FETCH_SIZE"] / reps,
            "write_bytes": 1024 * c["WRITE_SIZE"] / reps, "hbm_bytes": 1024 * (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) / reps,
            "l2_hit_rate": c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)} for n, c in kern.items()}
    out["per_frame_bytes"][key] = {p: 1024 * (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) / reps / batch for p, c in phase.items()}
    out["per_frame_detail"][key] = {p: {"fetch_bytes_raw": 1024 * c["FETCH_SIZE"] / reps / batch,
                                        "write_bytes": 1024 * c["WRITE_SIZE"] / reps / batch} for p, c in phase.items()}
json.dump(out, open("profiles/r02_pmc_hash.json", "w"), indent=1)
print(json.dumps(out["per_frame_bytes"], indent=1))
