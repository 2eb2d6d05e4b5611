"""in-kernel per-segment cycle sums of the flash self-attention loop (workgroup 0, one wave), head-major K / V as in the step"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library(); dev = "cuda"
buf = torch.zeros(8, dtype=torch.int64, device=dev)
names = ["wait + barrier", "LDS reads + QK^T", "softmax", "PV", "-", "-"]
for (B, H, L, d) in [(2, 8, 4096, 40), (2, 8, 1024, 80)]:
    qkv = torch.randn(B, L, 3 * H * d, device=dev).half(); C = H * d
    q, k, v = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
    k, v = (t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3) for t in (k, v))
    for var in ([0, 13, 17, 18] if d == 40 else [0]):
        for wave in ((0, 4) if var else (0,)):
            lib.dsc_debug_set_self_attn_variant(var); lib.dsc_debug_set_self_attn_stamp_wave(wave)
            lib.dsc_debug_set_self_attn_stamps(ctypes.c_void_p(buf.data_ptr()))
            for _ in range(5): ops.self_attention(q, k, v)
            torch.cuda.synchronize()
            lib.dsc_debug_set_self_attn_stamps(None)
            t = buf.cpu().tolist()[:6]; nt = L // 64
            print(f"L{L} d{d} variant {var} wave {wave}: per tile cycles:", " | ".join(f"{n} {x/nt:.0f}" for n, x in zip(names, t)), f"| total/tile {sum(t)/nt:.0f}")
lib.dsc_debug_set_self_attn_variant(0); lib.dsc_debug_set_self_attn_stamp_wave(0)
