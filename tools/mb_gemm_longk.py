"""long-K GEMMs of the 64x64 / 32x32 levels (the shapes hipBLASLt served until round 3) on gemm_tn_f16's tilings, in the step's
cache state (cold weights, warm activations: tools/mb_gemm_cold.py's protocol).  dsc_debug_set_gemm_stages encoding in dsc_hip.h."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import _lib, ops  # noqa: E402
lib = _lib.load_library()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def cold(fn, x, reps=9):
    xc = torch.empty_like(x)
    ts = []
    for r in range(reps + 1):
        flush.fill_(r)
        xc.copy_(x)
        e0.record(); fn(); e1.record(); e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


variants = {"default": 0, "64r/3st/load": 40643, "64r/3st/noload": 90643, "64r/2st": 90642, "128r/3st": 91283, "128r/2st": 91282,
            "128r/3st/load": 41283}
print("M N K | " + " | ".join(variants))
for (M, N, K) in [(8192, 320, 1280), (8192, 320, 960), (2048, 640, 2560), (2048, 640, 1920), (2048, 640, 1280), (2048, 640, 960), (8192, 320, 640)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).half(); w = (torch.randn(N, K, device="cuda") * 0.03).half()
    b = torch.randn(N, device="cuda").half(); r = torch.randn(M, N, device="cuda").half()
    row = []
    for name, code in variants.items():
        lib.dsc_debug_set_gemm_stages(code)
        row.append(cold(lambda: ops.linear(x, w, b, residual=r, prefer_kernel=True), x))
    lib.dsc_debug_set_gemm_stages(0)
    print(f"{M} {N} {K} | " + " | ".join(f"{t:6.1f}" for t in row), flush=True)
