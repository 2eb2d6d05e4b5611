"""microbench: region cross-attention / self-attention kernel time with launches captured in a HIP graph (no host
launch overhead): per-launch time = graph time / launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops
dev = "cuda"
def tm_graph(fn, n=40, reps=5):
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
which = sys.argv[1] if len(sys.argv) > 1 else "xattn"
if which == "xattn":
    for (Bc, H, L, d) in [(2, 8, 4096, 40), (2, 8, 1024, 80), (2, 8, 256, 160), (2, 8, 64, 160), (16, 8, 4096, 40)]:
        S, C = 77, H * d
        q = torch.randn(Bc, L, C, device=dev).half(); k = torch.randn(Bc, S, C, device=dev).half(); v = torch.randn(Bc, S, C, device=dev).half()
        w = torch.zeros(2, L, S, device=dev); w[:, : L // 3, 2:4] = 0.5
        sig = torch.tensor([7.0], device=dev)
        out = torch.empty(Bc, L, H, d, dtype=torch.half, device=dev)
        q4, k4, v4 = q.view(Bc, L, H, d), k.view(Bc, S, H, d), v.view(Bc, S, H, d)
        ng = Bc // 2
        f = lambda **kw: ops.region_xattn(q4, k4, v4, kw.pop("region", w), sig, layout="blhd", n_std_groups=ng, out=out, **kw)
        f()
        t_pair = tm_graph(lambda: f()); t_final = tm_graph(lambda: f(bias_is_final=True)); t_nobias = tm_graph(lambda: f(region=None))
        alg = Bc * (2 * (2 * L * C + 2 * S * C) + 4 * L * S)
        t_exit = tm_graph(lambda: f(region=None, debug_flags=8)); t_pro = tm_graph(lambda: f(region=None, debug_flags=16))
        print(f"   probes: exit-at-start {t_exit:.2f} us, prologue-only {t_pro:.2f} us")
        packed = ops.xattn_kv_pack(k4, v4); comp = ops.compress_region_table(w, pad_rows=True); comp = (comp[0], ops.pad_region_rows(comp[1]))
        fp = lambda **kw: ops.region_xattn_packed(q4, packed, S, kw.pop("region", comp), sig, n_std_groups=ng, out=out, **kw)
        fp()
        tp_pair = tm_graph(lambda: fp()); tp_fwd = tm_graph(lambda: fp(reuse_stats=True)); tp_nob = tm_graph(lambda: fp(region=None))
        tl_pair = tm_graph(lambda: fp(ref_fp16_rounding=False)); tl_fwd = tm_graph(lambda: fp(ref_fp16_rounding=False, reuse_stats=True)); tl_nob = tm_graph(lambda: fp(region=None, ref_fp16_rounding=False))
        for fl, nm in ((64, "tpw2"), (128, "tpw4")):
            tt = tm_graph(lambda: fp(ref_fp16_rounding=False, reuse_stats=True, debug_flags=fl))
            print(f"   PACKED fp32-score {nm}: fwd {tt:6.2f} us")
        print(f"   PACKED fp32-score: stats+fwd {tl_pair:6.2f} us  fwd {tl_fwd:6.2f} us ({alg/tl_fwd/1e3:6.0f} GB/s alg)  fwd(no bias) {tl_nob:6.2f} us")
        print(f"   PACKED: stats+fwd {tp_pair:6.2f} us  fwd {tp_fwd:6.2f} us ({alg/tp_fwd/1e3:6.0f} GB/s alg)  fwd(no bias) {tp_nob:6.2f} us")
        print(f"Bc{Bc} L{L} d{d}: stats+fwd {t_pair:6.2f} us ({alg/t_pair/1e3:6.0f} GB/s)  fwd(bias final, no std) {t_final:6.2f}  fwd(no bias) {t_nobias:6.2f} us", flush=True)
else:
    for (B, H, L, d) in [(2, 8, 4096, 40), (2, 8, 1024, 80), (2, 8, 256, 160), (2, 8, 64, 160)]:
        qkv = torch.randn(B, L, 3 * H * d, device=dev).half(); C = H * d
        q, k, v = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
        out = torch.empty(B, L, H, d, device=dev, dtype=torch.half)
        t1 = tm_graph(lambda: ops.self_attention(q, k, v, out=out))
        print(f"self-attn B{B} H{H} L{L} d{d}: {t1:8.2f} us ({4.0*B*H*L*L*d/t1/1e6:6.0f} TF)", flush=True)
x = torch.randn(2, 4096, 320, device=dev).half(); y = torch.randn(2, 4096, 320, device=dev).half()
print("torch add 5MB:", tm_graph(lambda: torch.add(x, y)), "us")
x = torch.randn(64, device=dev).half(); y = torch.randn(64, device=dev).half()
print("torch add 64 elems:", tm_graph(lambda: torch.add(x, y)), "us")
