"""microbench: MIOpen conv solver selection (graph-captured) for the SD1.5 3x3 shapes, channels-last fp16"""
import sys, os, torch, torch.nn.functional as F
dev = "cuda"
def tm_graph(fn, n=20, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3
torch.backends.cudnn.benchmark = (sys.argv[1] == "bench") if len(sys.argv) > 1 else False
print("cudnn.benchmark =", torch.backends.cudnn.benchmark, "MIOPEN_FIND_MODE =", os.environ.get("MIOPEN_FIND_MODE"))
for (B, cin, cout, hw, stride) in [(2, 320, 320, 64, 1), (2, 640, 640, 32, 1), (2, 1280, 1280, 16, 1), (2, 1280, 1280, 8, 1), (2, 640, 320, 64, 1), (2, 960, 320, 64, 1),
                                   (2, 2560, 1280, 16, 1), (2, 1920, 640, 32, 1), (2, 320, 320, 64, 2), (2, 320, 640, 32, 1)]:
    x = torch.randn(B, cin, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).half().contiguous(memory_format=torch.channels_last)
    t = tm_graph(lambda: F.conv2d(x, w, None, padding=1, stride=stride))
    fl = 2.0 * B * (hw // stride) ** 2 * cin * cout * 9
    print(f"conv3x3 B{B} {cin}->{cout} @{hw} s{stride}: {t:7.1f} us ({fl/t/1e6:5.0f} TF)", flush=True)
