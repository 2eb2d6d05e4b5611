"""PMC target: 20 launches of the flash self-attention kernel at the bench shape (B2 H8 L4096 d40) and of the 3x3 convolution"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops
dev = "cuda"
B, H, L, d = 2, 8, 4096, 40
C = H * d
qkv = torch.randn(B, L, 3 * C, device=dev).half()
q, k, v = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
k, v = (t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3) for t in (k, v))     # head-major, as dsc_linear_qkv_f16 writes them
q = q.contiguous()
x = torch.randn(2, 320, 64, 64, device=dev).half().contiguous(memory_format=torch.channels_last)
w = (torch.randn(320, 320, 3, 3, device=dev) / 54).half().contiguous(memory_format=torch.channels_last)
# the two convolution variants the step runs under the latency profile: 320 workgroups -> conv3x3_kernel<16, 3, 0> (320 -> 320 @ 64x64),
# 160 workgroups -> conv3x3_kernel<16, 9, 4> (640 -> 640 @ 32x32: nine-stage weight ring + four DMA-only loader waves)
x2 = torch.randn(2, 640, 32, 32, device=dev).half().contiguous(memory_format=torch.channels_last)
w2 = (torch.randn(640, 640, 3, 3, device=dev) / 76).half().contiguous(memory_format=torch.channels_last)
for _ in range(20):
    ops.self_attention(q, k, v)
    ops.conv3x3(x, w, None)
    ops.conv3x3(x2, w2, None)
torch.cuda.synchronize()
