"""microbench: our flash self-attention / region cross-attention vs torch SDPA at the SD1.5 shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from diffusionspatialcontrol_amd import ops
dev = "cuda"
def tm(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, L, d) in [(2, 8, 4096, 40), (2, 8, 1024, 80), (2, 8, 256, 160), (2, 8, 64, 160), (16, 8, 4096, 40), (2, 8, 9216, 40), (2, 10, 4096, 64)]:
    qkv = torch.randn(B, L, 3 * H * d, device=dev).half(); C = H * d
    q, k, v = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
    out = torch.empty(B, L, H, d, device=dev, dtype=torch.half)
    t1 = tm(lambda: ops.self_attention(q, k, v, out=out))
    qt, kt, vt = (x.transpose(1, 2).contiguous() for x in (q, k, v))
    t2 = tm(lambda: F.scaled_dot_product_attention(qt, kt, vt))
    fl = 4.0 * B * H * L * L * d
    print(f"self-attn B{B} H{H} L{L} d{d}: ours {t1:8.1f} us ({fl/t1/1e6:6.0f} TF)   torch SDPA {t2:8.1f} us ({fl/t2/1e6:6.0f} TF)", flush=True)
