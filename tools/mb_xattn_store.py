"""A/B: xp_fwd's output stores as 16-byte pieces (permlane32 swap between the two half-row lanes) against the 8-byte pieces
(debug flag 1024), forward launch alone (reuse_stats), hot graph-captured launches, all four SD1.5 levels + Bc = 16"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops
from mb_gemm_util import tm_graph
dev = "cuda"
for (Bc, H, L, d) in [(2, 8, 4096, 40), (2, 8, 1024, 80), (2, 8, 256, 160), (2, 8, 64, 160), (16, 8, 4096, 40), (2, 10, 4096, 64)]:
    S, C = 77, H * d
    q = torch.randn(Bc, L, C, device=dev).half(); k = torch.randn(Bc, S, C, device=dev).half(); v = torch.randn(Bc, S, C, device=dev).half()
    w = torch.zeros(2, L, S); w[:, : L // 3, 2:4] = 0.5
    sig = torch.tensor([7.0], device=dev)
    q4, k4, v4 = q.view(Bc, L, H, d), k.view(Bc, S, H, d), v.view(Bc, S, H, d)
    packed = ops.xattn_kv_pack(k4, v4)
    ids, rows = ops.compress_region_table(w, pad_rows=True)
    comp = (ids.to(dev), ops.pad_region_rows(rows).to(dev))
    out = torch.empty(Bc, L, H, d, dtype=torch.half, device=dev)
    f = lambda **kw: ops.region_xattn_packed(q4, packed, S, comp, sig, n_std_groups=Bc // 2, out=out, ref_fp16_rounding=False, **kw)
    a = f().clone(); b = f(debug_flags=1024).clone()
    res = []
    for _ in range(3):
        res.append((tm_graph(lambda: f(reuse_stats=True), n=40), tm_graph(lambda: f(reuse_stats=True, debug_flags=1024), n=40)))
    print(f"Bc{Bc} H{H} L{L} d{d}: equal {torch.equal(a, b)}; fwd 16-byte / 8-byte pieces: " + "  ".join(f"{x:.2f}/{y:.2f}" for x, y in res) + " us", flush=True)
