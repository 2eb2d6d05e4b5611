"""microbench: hipBLASLt GEMM + separate add vs dsc_linear_lt_f16 (bias epilogue + residual as beta*C), graph-captured"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from diffusionspatialcontrol_amd import ops
dev = "cuda"
def tm_graph(fn, n=30, reps=5):
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
for (M, N, K) in [(8192, 320, 1280), (2048, 640, 2560), (512, 1280, 1280), (512, 1280, 5120), (512, 1280, 2560), (128, 1280, 1280), (128, 1280, 5120), (128, 1280, 2560)]:
    x = torch.randn(M, K, device=dev).half(); w = torch.randn(N, K, device=dev).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half()
    ops.USE_DSC_GEMM = False
    ops.USE_LT_RESIDUAL = False
    t0 = tm_graph(lambda: ops.linear(x, w, b, residual=r))
    t00 = tm_graph(lambda: F.linear(x, w, b))
    ops.USE_LT_RESIDUAL = True
    t1 = tm_graph(lambda: ops.linear(x, w, b, residual=r))
    ops.USE_DSC_GEMM = True; ops.DSC_GEMM_MIN_ROWS = 1; ops.DSC_GEMM_MAX_K = 1 << 30
    t2 = tm_graph(lambda: ops.linear(x, w, b, residual=r))
    ops.DSC_GEMM_MIN_ROWS = 1024; ops.DSC_GEMM_MAX_K = 640
    print(f"M{M} N{N} K{K}: F.linear {t00:6.2f}  F.linear+add {t0:6.2f}  lt(bias+beta) {t1:6.2f}  gemm_tn {t2:6.2f} us", flush=True)
