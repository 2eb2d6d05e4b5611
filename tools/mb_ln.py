"""microbench: dsc_add_layernorm at the UNet's token shapes (graph-captured launches)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for rows, C in [(8192, 320), (2048, 640), (512, 1280), (128, 1280), (65536, 320)]:
    x = torch.randn(rows, C, device=dev).half(); a = torch.randn(rows, C, device=dev).half()
    g = torch.randn(C, device=dev).half(); b = torch.randn(C, device=dev).half()
    t = tm_graph(lambda: ops.add_layernorm(x, a, g, b))
    t0 = tm_graph(lambda: ops.add_layernorm(x, None, g, b))
    print(f"add_layernorm {rows}x{C}: with add {t:6.2f} us ({rows*C*8/t/1e6:5.2f} TB/s)   plain {t0:6.2f} us ({rows*C*4/t0/1e6:5.2f} TB/s)", flush=True)
