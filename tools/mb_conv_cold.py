"""3x3 convolutions of one SD1.5 UNet step in the step's cache state (weights from HBM, input just produced): split sweep.

The split-K cost model in conv3x3.hip (auto_splits) was fitted to back-to-back graph launches, which keep a layer's
weights (up to 29 MB) in the Infinity Cache; inside a step they come from HBM.  Every timed launch here follows a
512 MB fill and a copy of x (see tools/mb_gemm_cold.py).  Prints conv + reduce time per split count and the automatic choice.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import _lib, ops  # noqa: E402

lib = _lib.load_library()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def cold(fn, x, reps=7):
    xc = torch.empty_like(x)
    ts = []
    for r in range(reps + 1):
        flush.fill_(r)
        xc.copy_(x)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


# (Cin, Cout, hw, launches per step) at Bc = 2: ResNet conv1 / conv2, up-block conv1 on concatenations, up/down samplers
shapes = [(320, 320, 64, 11), (640, 320, 64, 2), (960, 320, 64, 1), (320, 640, 32, 1), (640, 640, 32, 10), (960, 640, 32, 1),
          (1280, 640, 32, 1), (1920, 640, 32, 1), (640, 1280, 16, 1), (1280, 1280, 16, 10), (1920, 1280, 16, 1),
          (2560, 1280, 16, 2), (1280, 1280, 8, 9), (2560, 1280, 8, 3)]
tot_auto = tot_best = 0.0
for cin, cout, hw, cnt in shapes:
    x = torch.randn(2, cin, hw, hw, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda").half()
    nc = cin // 64
    t_auto = cold(lambda: ops.conv3x3(x, w, b, splits=0), x)
    line = f"{cin:4d}->{cout:4d} @{hw:2d} x{cnt:2d}: auto {t_auto:6.1f} |"
    best = t_auto
    for s in range(1, nc + 1):
        if nc % s:
            continue
        t = cold(lambda: ops.conv3x3(x, w, b, splits=s), x)
        best = min(best, t)
        line += f" s{s}:{t:6.1f}"
    tot_auto += cnt * t_auto
    tot_best += cnt * best
    print(line, flush=True)
print(f"per step: automatic {tot_auto / 1e3:.3f} ms, per-shape best {tot_best / 1e3:.3f} ms")
