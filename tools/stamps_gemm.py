"""per-workgroup timeline of gemm_tn_f16 (dsc_debug_set_gemm_stamps) in the step's cache state (512 MB fill, then a copy of x):
start skew over the grid, prologue (entry -> first K tile published), K loop, epilogue (staging + stores retired), workgroups per CU"""
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library(); dev = "cuda"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
shapes = [(8192, 320, 320, True, False), (8192, 960, 320, False, False), (8192, 2560, 320, False, True), (8192, 320, 1280, True, False),
          (2048, 640, 640, True, False), (2048, 5120, 640, False, True), (2048, 640, 2560, True, False), (512, 1280, 1280, True, False)]
if os.environ.get("SHAPE"):
    v = os.environ["SHAPE"].split(","); shapes = [(int(v[0]), int(v[1]), int(v[2]), v[3] == "1", v[4] == "1")]
for (M, N, K, res, geglu) in shapes:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); xc = torch.empty_like(x)
    w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    nwg = 16384
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    ts = []
    for rep in range(4):
        flush.fill_(rep); xc.copy_(x); buf.zero_()
        if rep == 3: lib.dsc_debug_set_gemm_stamps(ctypes.c_void_p(buf.data_ptr()))
        e0.record()
        ops.linear(xc, w, b, residual=r, geglu=geglu, prefer_kernel=True)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    lib.dsc_debug_set_gemm_stamps(None)
    t = buf.cpu().numpy().reshape(nwg, 8)
    t = t[t[:, 0] != 0]
    if len(t) == 0:
        print(f"M{M} N{N} K{K}: no stamps (split-K or library path)"); continue
    t0 = t[:, 0].min()
    start, ls, le, end = [(t[:, i] - t0) / 100.0 for i in range(4)]       # us
    cu_key = [(int(v) >> 32, (int(v) >> 8) & 0xf, (int(v) >> 13) & 0x7, (int(v) >> 16) & 0xf) for v in t[:, 7]]
    per_cu = collections.Counter(cu_key)
    print(f"M{M} N{N} K{K} res={int(res)} geglu={int(geglu)}: event-timed {min(ts[1:3]):.1f} us (stamped launch {ts[3]:.1f}); {len(t)} workgroups on {len(per_cu)} CUs "
          f"(max {max(per_cu.values())}/CU); span {end.max():.1f} us; start mean {start.mean():.2f} max {start.max():.2f}; "
          f"prologue {np.mean(ls-start):.2f} (min {np.min(ls-start):.2f} max {np.max(ls-start):.2f}); loop {np.mean(le-ls):.2f} (max {np.max(le-ls):.2f}) "
          f"= {np.mean(t[:,5]) / max(K // 64, 1):.0f} cycles / K tile; epilogue {np.mean(end-le):.2f} (max {np.max(end-le):.2f}); last start {start.max():.2f}, first end {end.min():.2f}")
    # deciles of start and end over the workgroups
    qs = [0, 25, 50, 75, 100]
    print("      start pct " + " ".join(f"{np.percentile(start, q):.2f}" for q in qs) + " | end pct " + " ".join(f"{np.percentile(end, q):.2f}" for q in qs))
