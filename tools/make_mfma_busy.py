"""profiles/pmc_mfma_busy.json from the SQ-counter pass of tools/pmc_kernels.py (bench.py prints it as `mfma_busy_pct`):

    python tools/make_mfma_busy.py <counter_collection.csv> [tag]

Per kernel (averages over its dispatches):
  mfma_busy_cycles   SQ_VALU_MFMA_BUSY_CYCLES, summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_f16: guide,
                     per-instruction constants)
  kernel_cycles      SQ_BUSY_CYCLES / 32 - the counter is summed over the 32 shader engines, each busy for the kernel's
                     duration (checked against the kernel-trace duration x the ~1.8-2.1 GHz the chip holds under load)
  mfma_busy_pct      100 * mfma_busy_cycles / (kernel_cycles * 1024): the share of the CHIP's matrix-pipe cycles in use
  mfma_busy_pct_of_occupied_simds   the same over the SIMDs the grid can occupy (min(workgroups, 256) CUs x 4)
The JSON is stamped with the sha256 prefix of csrc/ so that bench.py can tell when the kernels have changed since."""
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

KEYS = [("xp_fwd<3", "xp_fwd"), ("xp_stats<3", "xp_stats"), ("self_attn_fwd<3", "self_attn_fwd"),
        ("conv3x3_kernel", None), ("gemm_tn_f16", None)]


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffusionspatialcontrol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    path = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    grids = {}
    kept, dur = [], collections.defaultdict(lambda: [0.0, 0])
    rdr = csv.DictReader(open(path))
    for r in rdr:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        if not any(k in name for k, _ in KEYS):
            continue
        wgs = (int(r["Grid_Size"]) // int(r["Workgroup_Size"])) if "Grid_Size" in r and r.get("Workgroup_Size") else None
        key = f"{name} [{wgs} workgroups]" if wgs else name
        grids[key] = wgs
        kept.append(r)
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            dd = dur[key]
            dd[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dd[1] += 1
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    rec = {"csrc_sha16": csrc_sha16(), "source_csv": f"profiles/{tag}_pmc_sq_counters_raw.csv", "kernels": {}}
    for key, ctrs in sorted(acc.items()):
        avg = {c: v[0] / v[1] for c, v in ctrs.items()}
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in avg or "SQ_BUSY_CYCLES" not in avg:
            continue
        kcyc = avg["SQ_BUSY_CYCLES"] / 32.0
        wgs = grids[key]
        simds = 4 * min(wgs, 256) if wgs else 1024
        e = {"dispatches": int(next(iter(ctrs.values()))[1]), "avg_duration_us_under_the_profiler": round(dur[key][0] / max(1, dur[key][1]), 2),
             "mfma_busy_cycles": round(avg["SQ_VALU_MFMA_BUSY_CYCLES"]),
             "kernel_cycles": round(kcyc), "mfma_busy_pct": round(100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (kcyc * 1024), 2),
             "mfma_busy_pct_of_occupied_simds": round(100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (kcyc * simds), 2)}
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_ACTIVE_INST_VALU"):
            if c in avg:
                e[c] = round(avg[c])
        rec["kernels"][key] = e
    json.dump(rec, open(os.path.join(OUT, "pmc_mfma_busy.json"), "w"), indent=1)
    with open(os.path.join(OUT, f"{tag}_pmc_sq_counters_raw.csv"), "w", newline="") as f:      # this package's kernels only
        wr = csv.DictWriter(f, fieldnames=rdr.fieldnames)
        wr.writeheader()
        wr.writerows(kept)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
