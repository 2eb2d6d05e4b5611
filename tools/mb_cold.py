"""microbench: how much of a weight-heavy kernel's in-pipeline time is the cold (HBM) first touch of its weights?
hot = weights resident; cold = 1 GiB streamed through the caches before each call; mall = cold, then the weights are read
once by a small copy kernel (any CU) right before the call - what a prefetcher could do"""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops
dev = "cuda"
scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def timed(fn, prep, n=12):
    ts = []
    for _ in range(n):
        prep()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
cases = []
x = torch.randn(512, 1280, device=dev).half(); w = torch.randn(1280, 1280, device=dev).half(); b = torch.randn(1280, device=dev).half()
cases.append(("hipBLASLt M512 N1280 K1280", lambda: F.linear(x, w, b), [w]))
w2 = torch.randn(10240, 1280, device=dev).half(); b2 = torch.randn(10240, device=dev).half()
cases.append(("hipBLASLt M512 N10240 K1280", lambda: F.linear(x, w2, b2), [w2]))
xc = torch.randn(2, 1280, 16, 16, device=dev).half().contiguous(memory_format=torch.channels_last)
wc = (torch.randn(1280, 1280, 3, 3, device=dev) / 100).half().contiguous(memory_format=torch.channels_last)
cases.append(("conv3x3 1280->1280 @16", lambda: ops.conv3x3(xc, wc, None), [wc]))
xc0 = torch.randn(2, 320, 64, 64, device=dev).half().contiguous(memory_format=torch.channels_last)
wc0 = (torch.randn(320, 320, 3, 3, device=dev) / 50).half().contiguous(memory_format=torch.channels_last)
cases.append(("conv3x3 320->320 @64", lambda: ops.conv3x3(xc0, wc0, None), [wc0]))
for name, fn, ws in cases:
    for _ in range(3): fn()
    hot = timed(fn, lambda: None)
    cold = timed(fn, lambda: scratch.fill_(1))
    def mall():
        scratch.fill_(1)
        for t in ws: t.sum()          # reads every line once (reduction kernel on some CUs)
    warm = timed(fn, mall)
    print(f"{name:32s} hot {hot:7.1f} us   cold {cold:7.1f} us   cold + weights re-read once {warm:7.1f} us   (single launches incl. ~launch latency)", flush=True)
