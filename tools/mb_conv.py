"""microbench: conv / GEMM layouts on the SD1.5 shapes (decides the activation layout of the UNet)"""
import torch, torch.nn.functional as F, time, sys
dev = "cuda"
def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(2, 320, 320, 64), (2, 640, 640, 32), (2, 1280, 1280, 16), (2, 1280, 1280, 8), (2, 640, 320, 64), (2, 2560, 1280, 16), (2, 960, 640, 32), (2,4,320,64), (2,320,4,64)]
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for (B, cin, cout, hw) in shapes:
        x = torch.randn(B, cin, hw, hw, device=dev).half(); w = torch.randn(cout, cin, 3, 3, device=dev).half(); b = torch.randn(cout, device=dev).half()
        xc = x.contiguous(memory_format=torch.channels_last); wc = w.contiguous(memory_format=torch.channels_last)
        t1 = tm(lambda: F.conv2d(x, w, b, padding=1)); t2 = tm(lambda: F.conv2d(xc, wc, b, padding=1))
        fl = 2 * B * hw * hw * cin * cout * 9
        y = F.conv2d(xc, wc, b, padding=1)
        print(f"bench={bench} conv3x3 B{B} {cin}->{cout} @{hw}: NCHW {t1:8.1f} us ({fl/t1/1e6:6.0f} TF)  NHWC {t2:8.1f} us ({fl/t2/1e6:6.0f} TF) out_cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)
# GEMMs of the transformer at L=4096
for (M, K, N) in [(8192, 320, 320), (8192, 320, 2560), (8192, 1280, 320), (2048, 640, 5120), (2048, 2560, 640), (512, 1280, 10240), (154, 768, 320), (8192, 320, 960)]:
    a = torch.randn(M, K, device=dev).half(); w = torch.randn(N, K, device=dev).half(); b = torch.randn(N, device=dev).half()
    t = tm(lambda: F.linear(a, w, b)); print(f"linear {M}x{K}x{N}: {t:7.1f} us  {2*M*K*N/t/1e6:6.0f} TF", flush=True)
q = torch.randn(2, 8, 4096, 40, device=dev).half()
t = tm(lambda: F.scaled_dot_product_attention(q, q, q)); print(f"sdpa L4096 d40: {t:.1f} us {4*2*8*4096*4096*40/t/1e6:.0f} TF")
q = torch.randn(2, 8, 1024, 80, device=dev).half()
t = tm(lambda: F.scaled_dot_product_attention(q, q, q)); print(f"sdpa L1024 d80: {t:.1f} us {4*2*8*1024*1024*80/t/1e6:.0f} TF")
q = torch.randn(2, 8, 4096, 64, device=dev).half()
t = tm(lambda: F.scaled_dot_product_attention(q, q, q)); print(f"sdpa L4096 d64: {t:.1f} us {4*2*8*4096*4096*64/t/1e6:.0f} TF")
x = torch.randn(2, 4096, 320, device=dev).half(); g = torch.ones(320, device=dev).half()
t = tm(lambda: F.layer_norm(x, (320,), g, g)); print(f"layer_norm 8192x320: {t:.1f} us")
