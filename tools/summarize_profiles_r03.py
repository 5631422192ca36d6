"""Turn the rocprofv3 outputs of tools/run_profiles_r03.sh (merged under gpurun_out/) into the committed kernel-stat
tables under profiles/r03_*. (Counter passes: the kernels are round 2's, profiles/r02_pmc_*.json still apply.)"""
import csv
import glob
import re

G = "gpurun_out"


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace(",", ";")[:110]


def kstats(tag, dst, cmd, note):
    src = glob.glob(f"{G}/r03_{tag}/**/p_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}  (MI355X; {note}); total GPU time {tot / 1e6:.1f} ms\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:45]:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    print(dst, f"{tot / 1e6:.1f} ms total")


kstats("f32_b8", "profiles/r03_bench_b8_fp32_kernel_stats.csv", "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       "default fp32 arithmetic = 6 split-bf16 plane products; 3 warm-up + 10 timed steps as one captured HIP graph each, a probe "
       "pass and 1 + 10 instrumented eager steps: 25 steps of kernels")
kstats("f32_b1", "profiles/r03_bench_b1_fp32_kernel_stats.csv", "bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants",
       "3 + 20 graph steps, a probe pass and 1 + 20 instrumented eager steps: 45 steps of kernels")
kstats("bf16_b8", "profiles/r03_bench_b8_bf16_kernel_stats.csv", "bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-variants",
       "25 steps of kernels as for fp32")
kstats("train_b4", "profiles/r03_train_b4_fp32_kernel_stats.csv", "bench.py --mode train --batch 4 --steps 5 --warmup 2",
       "7 training steps, fused BatchNorm kernels, every gradient written straight into the flat buffer (no accumulate adds), "
       "weight-gradient kernels on a second stream (the spin kernels are the stream probe)")
