"""per-kernel averages of a rocprofv3 --pmc counter_collection.csv:  python tools/summarize_pmc.py <csv> [name filter]"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:48]
    if flt and flt not in name:
        continue
    a = acc[(name, r["Counter_Name"])]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
print("kernel,counter,avg_per_dispatch,dispatches")
for (name, ctr), (tot, n) in sorted(acc.items()):
    print(f"{name},{ctr},{tot / n:.0f},{n}")
