"""profiles/in_step_kernels.json from a rocprofv3 kernel trace of bench.py (bench.py prints it as `roofline.in_step_us`):

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py --steps 6 --warmup 2 \\
        --no-cpu-baseline --no-batched-roofline --in-flight 1
    python tools/make_in_step.py <dir>/*/*_kernel_trace.csv [tag]

The whole-run `--stats` average of a kernel mixes two populations: the launches INSIDE the captured UNet step (operands just
produced by the previous kernel, weights cold, the launch ramping behind another kernel) and the isolated back-to-back
launches of bench.py's roofline legs (what the line's live HIP-event figures time).  This tool separates them: `in_step` =
the last 20 UNet steps of the trace (the sampler's `step_kernel` delimits them, as in tools/step_breakdown.py), `hot_leg` =
the launches after the last sampler step (the roofline legs run after the generations).  Stamped with the sha256 prefix of
csrc/: bench.py drops the figures when the kernels have changed since."""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# where the stamped JSON / CSV copies go: profiles/ here; on the GPU box a directory under gpurun_out/ (only that comes back)
OUT = os.environ.get("DSC_PROFILES_DIR") or os.path.join(ROOT, "profiles")
# (substring of the demangled name, grid filter on Grid_Size_X or None, key on the bench line)
WANT = [("xp_fwd<3,", None, "xp_fwd"), ("xp_stats<3,", None, "xp_stats"), ("self_attn_fwd<3, 8", None, "self_attn_fwd"),
        ("conv3x3_kernel<16, 3, 0>", "81920", "conv3x3_320_320_64x64")]


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffusionspatialcontrol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    path = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "step_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 22, "fewer than 22 sampler steps in the trace"
    lo, hi = marks[-21], marks[-1]

    def key_of(r):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        for sub, grid, key in WANT:
            if sub in name and (grid is None or r["Grid_Size_X"] == grid):
                return key
        return None

    acc = collections.defaultdict(lambda: {"in_step": [0.0, 0], "hot_leg": [0.0, 0], "whole_run": [0.0, 0]})
    for i, r in enumerate(rows):
        k = key_of(r)
        if k is None:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for pop, cond in (("in_step", lo < i <= hi), ("hot_leg", i > hi), ("whole_run", True)):
            if cond:
                acc[k][pop][0] += d
                acc[k][pop][1] += 1
    step_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo + 1:hi + 1]) / 20.0
    rec = {"csrc_sha16": csrc_sha16(), "from": f"profiles/{tag}_bench_kernel_stats.csv / {tag}_step_breakdown.txt (same trace)",
           "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "
                      "--no-batched-roofline --no-coalesced --in-flight 1",
           "step_kernel_time_ms": round(step_ns / 1e6, 4), "kernels_per_step": (hi - lo) / 20.0, "kernels": {}}
    for k, pops in acc.items():
        rec["kernels"][k] = {p: {"avg_us": round(v[0] / v[1], 3), "launches": v[1]} for p, v in pops.items() if v[1]}
    json.dump(rec, open(os.path.join(OUT, "in_step_kernels.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
