"""microbench: the K-split eight-wave convolution (dsc_debug_set_conv_ring(502)) against the four-wave forms (500), hot
graph-captured launches, ring depth 9 / 3 - the SD1.5 shapes whose grids have <= 320 workgroups"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops, _lib
from mb_gemm_util import tm_graph
lib = _lib.load_library()
dev = "cuda"
shapes = [(2, 320, 320, 64), (2, 640, 640, 32), (2, 1280, 640, 32), (2, 1920, 640, 32), (2, 960, 640, 32), (2, 320, 640, 32), (2, 1280, 1280, 16), (2, 2560, 1280, 16),
          (2, 1280, 1280, 8), (2, 640, 320, 64), (16, 320, 320, 64), (16, 640, 640, 32)]
for (B, cin, cout, hw) in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * hw * hw * cin * cout * 9
    line = f"conv3x3 B{B} {cin:4d}->{cout:4d} @{hw:2d}:"
    for k2, ring, gb, name in ((500, 9, 601, "k1 r9"), (500, 3, 601, "k1 r3"), (502, 9, 600, "k2 r9 tap"), (502, 9, 601, "k2 r9 row"), (502, 9, 602, "k2 r9 row+ld")):
        lib.dsc_debug_set_conv_ring(k2); lib.dsc_debug_set_conv_ring(ring); lib.dsc_debug_set_conv_ring(gb)
        t = tm_graph(lambda: ops.conv3x3(x, w, None))
        line += f"  {name}: {t:6.1f} ({fl/t/1e6:4.0f} TF)"
    lib.dsc_debug_set_conv_ring(0); lib.dsc_debug_set_conv_ring(500); lib.dsc_debug_set_conv_ring(601)
    print(line, flush=True)
