"""Two generations in flight: what each kernel family costs when two streams share the chip.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batched-roofline
    python tools/overlap_breakdown.py OUT/*/*_kernel_trace.csv

Takes the trace's last window in which two queues run sampler steps concurrently (the in-flight leg), and per family prints the
summed kernel durations per step under overlap beside the same family's per-step sum in the one-at-a-time leg earlier in the
same trace: the ratio says which kernels pay for sharing the chip.  Wall time per step of the window = chip time per step.
"""
import collections
import csv
import gzip
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from step_breakdown import family  # noqa: E402


def main(path):
    rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    marks = [(int(r["End_Timestamp"]), r[qkey]) for r in rows if "step_kernel" in r["Kernel_Name"]]
    # the in-flight leg: the trailing run of sampler steps in which the queue alternates / differs from the first legs' queue
    queues = collections.Counter(q for _, q in marks)
    if len(queues) < 2:
        print("only one queue ran sampler steps: not an in-flight trace")
        return
    # window: from the first step of the SECOND most recent queue's last burst to the last step
    last_two = [q for q, _ in queues.most_common()]
    # walk back while both queues appear within any 40 consecutive marks
    both = [len({q for _, q in marks[max(0, k - 10):k + 10]}) >= 2 for k in range(len(marks))]
    hi = max(k for k in range(len(marks)) if both[k])
    lo = hi
    while lo > 0 and both[lo - 1]:
        lo -= 1
    lo, hi = lo + 25, hi - 25                                  # the steady part: not the first / last generation's ramp
    i = lo
    t_begin, t_end = marks[lo][0], marks[hi][0]
    steps = sum(1 for t, _ in marks if t_begin < t <= t_end)
    fam = collections.defaultdict(float)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t_begin and e <= t_end:
            fam[family(r["Kernel_Name"])] += e - s
    # one-at-a-time reference: 20 steps right before the window on a single queue
    ref = collections.defaultdict(float)
    j = i
    ref_marks = marks[max(0, j - 60):j - 20]
    if len({q for _, q in ref_marks}) == 1 and len(ref_marks) >= 21:
        r0, r1 = ref_marks[0][0], ref_marks[20][0]
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s >= r0 and e <= r1:
                ref[family(r["Kernel_Name"])] += (e - s) / 20.0
    wall = (t_end - t_begin) / steps
    tot = sum(fam.values()) / steps
    print(f"{path}: {steps} steps on {len(queues)} queues; chip time {wall / 1e6:.3f} ms / step, summed kernel time {tot / 1e6:.3f} ms / step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        one = ref.get(k)
        print(f"   {k:12s} {v / steps / 1e6:6.3f} ms / step overlapped" + (f"   {one / 1e6:6.3f} alone   x{v / steps / one:4.2f}" if one else ""))


if __name__ == "__main__":
    main(sys.argv[1])
