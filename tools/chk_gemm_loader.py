"""check: the loader-wave GEMM kernels against torch (fp32 accumulate), and their time beside the plain kernels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from diffusionspatialcontrol_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mb_gemm_util import tm_graph
lib = _lib.load_library()
dev = "cuda"
torch.manual_seed(0)
ops.DSC_GEMM_MIN_ROWS = 1; ops.DSC_GEMM_MAX_K = 1 << 30
for (M, N, K, geglu) in [(8192, 320, 320, 0), (8192, 2560, 320, 1), (8192, 320, 1280, 0), (2048, 640, 640, 0), (2048, 5120, 640, 1), (2048, 640, 2560, 0),
                         (512, 1280, 1280, 0), (512, 10240, 1280, 1), (512, 1280, 5120, 0), (130, 1280, 1280, 0), (77, 320, 768, 0)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half()
    ref = F.linear(x.float(), w.float(), b.float())
    ref = (ref[:, :N // 2] * F.gelu(ref[:, N // 2:])) if geglu else ref + r.float()
    line = []
    for stg in (90003, 40003, 90643, 40643):     # 9....: never loader waves, 4....: always
        if geglu and stg in (90643, 40643):
            continue
        lib.dsc_debug_set_gemm_stages(stg)
        fn = (lambda: ops.linear(x, w, b, geglu=True)) if geglu else (lambda: ops.linear(x, w, b, residual=r))
        err = (fn().float() - ref).abs().max().item()
        best = min(tm_graph(fn) for _ in range(3))
        line.append(f"{stg}: {best:7.2f} us err {err:.2e}")
    lib.dsc_debug_set_gemm_stages(0)
    print(f"M{M} N{N} K{K} {'geglu' if geglu else 'bias+res'}: " + "   ".join(line), flush=True)
