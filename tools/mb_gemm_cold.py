"""Linear layers of one SD1.5 UNet step, timed one launch at a time in the step's cache state.

Inside a step every weight comes from HBM (1.7 GB of weights pass between two uses of one) while the activations were
produced a moment ago.  Back-to-back repeats of one GEMM keep its weight in the Infinity Cache and rank kernels wrongly,
so every timed launch here follows a 512 MB fill (evicts everything) and a copy of x (brings the activations back).
Compares, per distinct (M, N, K, epilogue) of the step: the library GEMM (dsc_linear_lt_f16, algorithm as selected by
DSC_LT_TUNE) and the hand-written gemm_tn_f16 (dsc_linear_f16) wherever its tile constraints allow - i.e. it checks the
dispatch thresholds in ops.linear (DSC_GEMM_MIN_ROWS / DSC_GEMM_MAX_K).
"""
import collections
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import _lib, ops  # noqa: E402
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig  # noqa: E402

shapes = collections.OrderedDict()
_linear = ops.linear


def logged(x, weight, bias=None, residual=None, geglu=False):
    N, K = weight.shape
    M = x.numel() // K
    if M >= 8:
        key = (M, N, K, residual is not None, bool(geglu))
        shapes[key] = shapes.get(key, 0) + 1
    return _linear(x, weight, bias, residual=residual, geglu=geglu)


ops.linear = logged
torch.manual_seed(0)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
ops.USE_LN_FOLD = False            # log the plain linears of the folded blocks too
with torch.no_grad():
    unet(torch.randn(2, 4, 64, 64).half().cuda(), torch.tensor([500.0, 500.0]).cuda(),
         torch.randn(2, 77, 768).half().cuda())
ops.linear = _linear

lib = _lib.load_library()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def cold(fn, x, reps=7):
    xc = torch.empty_like(x)
    ts = []
    for r in range(reps + 1):
        flush.fill_(r)
        xc.copy_(x)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


print(f"{'M':>6} {'N':>6} {'K':>5} res geglu  n/step |  library  gemm_tn  (us, cold) | dispatch now")
tot_now = tot_best = 0.0
for (M, N, K, res, geglu), cnt in shapes.items():
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, device="cuda") * 0.03).half()
    b = torch.randn(N, device="cuda").half()
    r = torch.randn(M, N, device="cuda").half() if res else None
    n_out = N // 2 if geglu else N
    out = torch.empty(M, n_out, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()

    def run_lt():
        y = torch.empty(M, N, device="cuda", dtype=torch.float16) if geglu else out
        rc = lib.dsc_linear_lt_f16(p(x), p(w), p(b), p(r), p(y), M, N, K, K, N if res else 0, N, 0, s)
        assert rc == 0, rc
        if geglu:
            ops.geglu(y)

    def run_tn():
        rc = lib.dsc_linear_f16(p(x), p(w), p(b), p(r), p(out), M, N, K, K, N if res else 0, n_out, 1 if geglu else 0, 0, s)
        assert rc == 0, rc

    t_lt = cold(run_lt, x)
    tn_ok = K % 64 == 0 and N % 64 == 0 and (not geglu or (N // 2) % 32 == 0)
    t_tn = cold(run_tn, x) if tn_ok else float("nan")
    t_sk = {}
    if tn_ok and not geglu and K >= 1280:                     # split-K (dsc_linear_splitk_f16): auto and forced split counts
        ref = torch.nn.functional.linear(x.float(), w.float(), b.float()) + (r.float() if res else 0)
        for sp in (0, 2, 4, 8):
            t_sk[sp] = cold(lambda: ops.linear_splitk(x, w, b, r, splits=sp), x)
        got = ops.linear_splitk(x, w, b, r).float()
        assert (got - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-2, (got - ref).abs().max().item()
    now = "gemm_tn" if ops.linear_kernel_covers(M, N, K, torch.float16, geglu) else "library"
    t_now = t_tn if now == "gemm_tn" else t_lt
    t_best = min(t_lt, t_tn) if tn_ok else t_lt
    tot_now += cnt * t_now
    tot_best += cnt * t_best
    flag = "" if t_now <= t_best * 1.03 else "   <-- other path faster"
    print(f"{M:6d} {N:6d} {K:5d} {int(res):3d} {int(geglu):5d} {cnt:7d} | {t_lt:8.1f} {t_tn:8.1f}               | {now}{flag}  splitK auto/2/4/8: {' '.join(f'{v:5.1f}' for v in t_sk.values())}", flush=True)
print(f"per step: current dispatch {tot_now / 1e3:.3f} ms, per-shape best {tot_best / 1e3:.3f} ms")
