"""microbench: dsc_groupnorm_silu_nhwc kernel choices per UNet shape (graph-captured launches)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library(); dev = "cuda"
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
for (B, C, hw) in [(2, 320, 64), (2, 640, 64), (2, 960, 64), (2, 320, 32), (2, 640, 32), (2, 960, 32), (2, 1280, 32), (2, 1920, 32),
                   (2, 640, 16), (2, 1280, 16), (2, 1920, 16), (2, 2560, 16), (2, 1280, 8), (2, 2560, 8)]:
    x = torch.randn(B, C, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    g = torch.randn(C, device=dev).half(); b = torch.randn(C, device=dev).half()
    add = torch.randn(B, C, device=dev).half()
    line = f"GN B{B} C{C:4d} @{hw:2d}:"
    for mode in (0, 2, 4):
        lib.dsc_debug_set_gn_mode(mode)
        t = tm_graph(lambda: ops.groupnorm_silu_nhwc(x, 32, g, b, 1e-5, True, add=add))
        line += f"  mode{mode} {t:6.2f} us"
    lib.dsc_debug_set_gn_mode(0)
    print(line + f"   ({B*C*hw*hw*4/1e6:.1f} MB r+w)", flush=True)
