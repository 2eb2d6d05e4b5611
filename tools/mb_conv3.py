"""microbench: dsc_conv3x3_nhwc_f16 vs MIOpen (graph-captured launches) on the SD1.5 3x3 shapes; split sweep"""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library()
dev = "cuda"
torch.backends.cudnn.benchmark = True
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
shapes = [(2, 320, 320, 64), (2, 640, 320, 64), (2, 960, 320, 64), (2, 640, 640, 64), (2, 320, 640, 32), (2, 640, 640, 32),
          (2, 960, 640, 32), (2, 1280, 640, 32), (2, 1920, 640, 32), (2, 1280, 1280, 32), (2, 640, 1280, 16), (2, 1280, 1280, 16),
          (2, 1920, 1280, 16), (2, 2560, 1280, 16), (2, 1280, 1280, 8), (2, 2560, 1280, 8), (16, 320, 320, 64), (16, 1280, 1280, 8)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for (B, cin, cout, hw) in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * hw * hw * cin * cout * 9
    t0 = tm_graph(lambda: F.conv2d(x, w, None, padding=1))
    line = f"conv3x3 B{B} {cin:4d}->{cout:4d} @{hw:2d}: miopen {t0:7.1f} us ({fl/t0/1e6:4.0f} TF) | dsc"
    nc = cin // 64
    for s in [0]:
        t = tm_graph(lambda: ops.conv3x3(x, w, None, splits=s))
        line += f" s{s}:{t:6.1f}({fl/t/1e6:4.0f})"
    for ring in (3, 9):
        lib.dsc_debug_set_conv_ring(ring)
        t = tm_graph(lambda: ops.conv3x3(x, w, None))
        line += f" r{ring}:{t:6.1f}"
    lib.dsc_debug_set_conv_ring(0)
    ref = F.conv2d(x, w, None, padding=1)
    err = (ops.conv3x3(x, w, None) - ref).abs().max().item()
    print(line + f" | maxdiff {err:.2e}", flush=True)
