"""hot timing of the many-row GEMM shapes of 8 images per generation: the library's kernels against torch's (rocBLAS / hipBLASLt)
plain matmul as a yardstick for what the machine gives these shapes (tools/ubench/gemm_big.hip is the persistent-kernel probe)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops
dev = "cuda"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def hot(fn, reps=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(ts)
for (M, N, K, geglu) in [(65536, 2560, 320, True), (16384, 5120, 640, True), (4096, 10240, 1280, True), (65536, 640, 320, False), (65536, 320, 320, False),
                         (16384, 640, 640, False), (4096, 1280, 1280, False), (65536, 1280, 1280, False), (65536, 320, 1280, False), (16384, 640, 2560, False),
                         (65536, 3840, 320, False), (65536, 960, 320, False)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    t_own = hot(lambda: ops.linear(x, w, b, geglu=geglu, prefer_kernel=True))
    t_lib = hot(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K} geglu={int(geglu)}: own {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF/s   torch linear (no GELU) {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF/s", flush=True)
