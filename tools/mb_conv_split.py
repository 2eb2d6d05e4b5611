import sys, os
sys.path.insert(0, "/root/repo")
import torch
from diffusionspatialcontrol_amd import ops
dev = "cuda"
def tm_graph(fn, n=20, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3
for (B, cin, cout, hw) in [(2, 640, 640, 32), (2, 1280, 640, 32), (2, 1920, 640, 32), (2, 320, 320, 64), (2, 640, 320, 64)]:
    x = torch.randn(B, cin, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev).half()
    res = {}
    for rnd in range(3):
        for sp in (1, 2, 3, 5):
            if (cin // 64) % sp: continue
            t = tm_graph(lambda: ops.conv3x3(x, w, b, splits=sp))
            res[sp] = min(res.get(sp, 1e9), t)
    print(f"B{B} {cin}->{cout} @{hw}: " + "  ".join(f"splits {k}: {v:.1f} us" for k, v in res.items()), flush=True)
