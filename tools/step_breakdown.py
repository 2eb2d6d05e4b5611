"""Steady-state breakdown of one UNet step from a rocprofv3 kernel trace of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python bench.py --steps 6 --warmup 2 \\
        --no-cpu-baseline --no-batched-roofline --in-flight 1
    python tools/step_breakdown.py gpurun_out/prof/*/*_kernel_trace.csv

The fused sampler kernel (`step_kernel`, one per step) delimits the steps; the last 20 steps of the run are averaged,
so warm-up, graph capture, algorithm timing (linear_lt.hip) and the roofline launches do not enter - the whole-run
`--stats` averages under profiles/ include them.  This is the A/B instrument when boxes differ by +-5 % in images/s.
"""
import collections
import csv
import re
import sys

FAMILIES = [("conv3x3_kernel", "conv3x3"), ("conv3x3_reduce", "conv_reduce"), ("gemm_splitk_reduce", "gemm_reduce"), ("gemm_tn", "gemm_tn"), ("self_attn", "self_attn"),
            ("gn_nhwc", "groupnorm"), ("xp_", "xattn"), ("CatArray", "cat"), ("add_ln", "add_ln")]


def family(name):
    if name.startswith("Cijk"):
        return "hipblaslt"
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def main(paths, verbose):
    for path in paths:
        rows = list(csv.DictReader(open(path)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        marks = [i for i, r in enumerate(rows) if "step_kernel" in r["Kernel_Name"]]
        if len(marks) < 22:
            print(path, ": fewer than 22 sampler steps in the trace")
            continue
        fam = collections.defaultdict(lambda: [0, 0])
        lib = collections.defaultdict(lambda: [0, 0])
        per = collections.defaultdict(lambda: [0, 0])
        wall = n = 0
        for a, b in zip(marks[-21:-1], marks[-20:]):
            n += 1
            wall += int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])
            for r in rows[a + 1:b + 1]:
                d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                f = fam[family(r["Kernel_Name"])]
                f[0] += d
                f[1] += 1
                short = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                short = re.sub(r"\(.*", "", short)[:60] if not short.startswith("Cijk") else "Cijk " + re.search(r"MT\d+x\d+x\d+", short).group(0)
                pk = per[short + " grid " + r["Grid_Size_X"] + "x" + r.get("Grid_Size_Y", "1") + "x" + r.get("Grid_Size_Z", "1")]
                pk[0] += d
                pk[1] += 1
                if r["Kernel_Name"].startswith("Cijk"):
                    k = lib[re.search(r"MT\d+x\d+x\d+", r["Kernel_Name"]).group(0) + " grid " + r["Grid_Size_X"]]
                    k[0] += d
                    k[1] += 1
        total = sum(v[0] for v in fam.values())
        print("%s: %d steps, kernel time %.3f ms / step, wall %.3f ms / step, %.0f kernels / step"
              % (path, n, total / n / 1e6, wall / n / 1e6, sum(v[1] for v in fam.values()) / n))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            print("   %-12s %6.3f ms %6.1f launches" % (k, v[0] / n / 1e6, v[1] / n))
        if verbose > 1:
            for k, v in sorted(per.items(), key=lambda kv: -kv[1][0]):
                print("      %-90s %7.1f us / step %5.1f x %6.1f us" % (k, v[0] / n / 1e3, v[1] / n, v[0] / v[1] / 1e3))
        elif verbose:
            for k, v in sorted(lib.items(), key=lambda kv: -kv[1][0]):
                print("      %-30s %7.1f us / step %5.1f x %6.1f us" % (k, v[0] / n / 1e3, v[1] / n, v[0] / v[1] / 1e3))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a not in ("-v", "-vv")]
    main(args, 2 if "-vv" in sys.argv else (1 if "-v" in sys.argv else 0))
