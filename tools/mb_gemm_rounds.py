"""cold-protocol timing of one GEMM shape under forced ring depths / tile heights (dsc_debug_set_gemm_stages encoding: stages + 10 * bm):
does a launch whose workgroups do not fill a whole number of residency rounds prefer the other ring depth?"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library(); dev = "cuda"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def cold(fn, x, reps=9):
    xc = torch.empty_like(x); ts = []
    for r in range(reps + 1):
        flush.fill_(r); xc.copy_(x)
        e0.record(); fn(xc); e1.record(); e1.synchronize()
        if r: ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for (M, N, K, res, geglu) in [(8192, 960, 320, False, False), (8192, 320, 1280, True, False), (8192, 2560, 320, False, True), (2048, 1920, 640, False, False), (2048, 640, 2560, True, False), (2048, 5120, 640, False, True)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    out = []
    for code, name in ((0, "default"), (2, "2 stages"), (3, "3 stages"), (643, "64 rows, 3 stages"), (1282, "128 rows, 2 stages"), (1283, "128 rows, 3 stages")):
        lib.dsc_debug_set_gemm_stages(code)
        try:
            t = min(cold(lambda xc: ops.linear(xc, w, b, residual=r, geglu=geglu, prefer_kernel=True), x) for _ in range(2))
            out.append(f"{name}: {t:.1f}")
        except Exception as e:
            out.append(f"{name}: -")
    lib.dsc_debug_set_gemm_stages(0)
    print(f"M{M} N{N} K{K} res={int(res)} geglu={int(geglu)}: " + "  ".join(out), flush=True)
