"""microbench: dsc_linear_f16 vs hipBLASLt (graph-captured launches)"""
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
for (M, N, K, geglu) in [(8192, 320, 320, 0), (8192, 960, 320, 0), (8192, 2560, 320, 1), (8192, 320, 1280, 0), (2048, 640, 640, 0), (2048, 1920, 640, 0),
                         (2048, 5120, 640, 1), (2048, 640, 2560, 0), (512, 1280, 1280, 0), (512, 3840, 1280, 0), (512, 10240, 1280, 1), (512, 1280, 5120, 0),
                         (128, 1280, 1280, 0), (128, 10240, 1280, 1), (128, 1280, 5120, 0)]:
    x = torch.randn(M, K, device=dev).half(); w = torch.randn(N, K, device=dev).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half()
    ops.DSC_GEMM_MIN_ROWS = 1; ops.DSC_GEMM_MAX_K = 1 << 30
    from diffusionspatialcontrol_amd import _lib
    lib = _lib.load_library()
    res = {}
    # dsc_debug_set_gemm_stages encoding (include/dsc_hip.h): 9xxxx = no loader waves, 1xxxxx = no 128-column tiles
    for stg in (190003, 190002, 190643, 190642, 190003, 190002, 190643, 190642):
        lib.dsc_debug_set_gemm_stages(stg)
        t = tm_graph(lambda: ops.linear(x, w, b, geglu=True)) if geglu else tm_graph(lambda: ops.linear(x, w, b, residual=r))
        res[stg] = min(res.get(stg, 1e9), t)
    lib.dsc_debug_set_gemm_stages(0)
    t2 = tm_graph(lambda: ops.geglu(F.linear(x, w, b))) if geglu else tm_graph(lambda: F.linear(x, w, b) + r)
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K} {'geglu' if geglu else 'bias+res'}: ours 128x64 tiles, 3 / 2 stages {res[190003]:7.2f} / {res[190002]:7.2f} us   64x64 tiles {res[190643]:7.2f} / {res[190642]:7.2f} us   hipBLASLt+epilogue kernels {t2:7.2f} us ({fl/t2/1e6:5.0f} TF)", flush=True)
