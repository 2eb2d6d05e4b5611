"""PMC target (LDS-side counters): 20 launches each of conv3x3 640 -> 640 @ 32x32 (160 workgroups), 320 -> 320 @ 64x64 (320), and
gemm_tn_f16 M = 8192 N = 320 K = 320 / the 64x64 GEGLU projection, Bc = 2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops
dev, N = "cuda", 20
g = torch.Generator().manual_seed(3)
cl = torch.channels_last
for (cin, hw) in ((640, 32), (320, 64)):
    x = torch.randn(2, cin, hw, hw, generator=g).half().to(dev).contiguous(memory_format=cl)
    wt = (torch.randn(cin, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).half().to(dev).contiguous(memory_format=cl)
    for _ in range(N):
        ops.conv3x3(x, wt, None)
    torch.cuda.synchronize()
for (M, K, Nn, geglu) in ((8192, 320, 320, False), (8192, 320, 2560, True)):
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    wt = (torch.randn(Nn, K, generator=g) / K ** 0.5).half().to(dev)
    b = torch.zeros(Nn).half().to(dev)
    for _ in range(N):
        ops.linear(x, wt, b, geglu=geglu, prefer_kernel=True)
    torch.cuda.synchronize()
