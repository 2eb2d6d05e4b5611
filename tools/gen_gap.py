"""Between two generations of the one-at-a-time leg: what the GPU does from the last sampler step of generation i to the first
kernel of generation i+1's first UNet step (a rocprofv3 kernel trace of bench.py --in-flight 1): kernels, busy time, idle time.

    python tools/gen_gap.py <kernel_trace.csv> [steps per generation = 25]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 25
marks = [i for i, r in enumerate(rows) if "step_kernel" in r["Kernel_Name"]]
# the last steps of the trace are the timed one-at-a-time leg; walk back over whole generations
gaps = []
for g in range(1, 5):
    last = marks[-1 - g * per]                               # last sampler step of a generation
    nxt_first_step_end = marks[-g * per]                     # first sampler step of the next generation
    # kernels of the next generation's first UNet step: the 353 before that marker
    k0 = nxt_first_step_end - 353
    between = rows[last + 1:k0]
    t_end = int(rows[last]["End_Timestamp"])
    t_start = int(rows[k0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in between)
    names = collections.Counter(re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]))[:50] for r in between)
    gaps.append((t_start - t_end, busy, len(between)))
    if g == 1:
        print("kernels between two generations:", len(between))
        for n, c in names.most_common(14):
            tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in between if n in re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]))
            print(f"   {c:4d} x {n:52s} {tot / 1e3:8.1f} us")
for span, busy, n in gaps:
    print(f"last step -> next generation's first kernel: {span / 1e6:.3f} ms, of which kernels {busy / 1e6:.3f} ms in {n} launches, idle {(span - busy) / 1e6:.3f} ms")
