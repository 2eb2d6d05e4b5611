"""PMC target: the plain K = 320 GEMM at 8 images per generation (M = 65536, N = 320) and at batch 1 (M = 8192), 10 launches each,
inputs rewritten between launches (as in the step: the activations were just produced)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops
g = torch.Generator().manual_seed(1)
for M in (65536, 8192):
    x = torch.randn(M, 320, generator=g).half().cuda()
    w = (torch.randn(320, 320, generator=g) / math.sqrt(320)).half().cuda()
    b = torch.zeros(320).half().cuda()
    r = torch.randn(M, 320, generator=g).half().cuda()
    for _ in range(10):
        x.mul_(1.0)                                   # rewrites x: the GEMM's input comes from a producer, not from its own last read
        ops.linear(x, w, b, residual=r, prefer_kernel=True)
    torch.cuda.synchronize()
