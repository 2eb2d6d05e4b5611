"""check: the Gram-moment identity for the region cross-attention's std - external (sum a, sum a^2) partials (ops.gram_stats_reference,
what dsc_linear_q_gram_f16's epilogue emits) against the statistics kernel's own, through the forward kernel"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import ops
from oracle import region_attention as ra
from mb_gemm_util import tm_graph
dev = "cuda"
g = torch.Generator().manual_seed(5)
for (Bt, H, L, d, ng, shared) in [(2, 8, 4096, 40, 1, False), (2, 8, 1024, 80, 1, False), (4, 8, 1024, 40, 2, False), (2, 8, 4096, 40, 1, True), (16, 8, 1024, 80, 8, False)]:
    S, C = 77, H * d
    Bq = Bt // 2 if shared else Bt
    q = (torch.randn(Bq, L, C, generator=g) * 1.3 + 0.1).half().to(dev)
    k = (torch.randn(Bt, S, C, generator=g) * 2.0 + 0.3).half().to(dev)
    v = torch.randn(Bt, S, C, generator=g).half().to(dev)
    w = torch.zeros(Bt, L, S); w[:, : L // 3, 2:4] = 0.5
    sig = torch.tensor([7.0], device=dev)
    k4, v4 = k.view(Bt, S, H, d), v.view(Bt, S, H, d)
    q4 = q.view(Bq, L, H, d)
    qfull = q4.repeat(2, 1, 1, 1) if shared else q4
    packed = ops.xattn_kv_pack(k4, v4)
    ids, rows = ops.compress_region_table(w, pad_rows=True)
    comp = (ids.to(dev), ops.pad_region_rows(rows).to(dev))
    a = ops.region_xattn_packed(qfull, packed, S, comp, sig, n_std_groups=ng, ref_fp16_rounding=False)
    gram = ops.xattn_gram_pack(k4)
    ext = ra.score_moment_partials(q4, k4, ng)
    b = ops.region_xattn_packed(qfull, packed, S, comp, sig, n_std_groups=ng, ref_fp16_rounding=False, ext_stats=ext)
    std_k = ops.region_xattn_std(qfull, k4, layout="blhd", n_std_groups=ng, ref_fp16_rounding=False)
    n = (Bt // ng) * H * L * S
    s = ext.sum(dim=1)
    std_g = torch.sqrt((s[:, 1] - s[:, 0] ** 2 / n) / (n - 1)).float()
    print(f"Bt{Bt} L{L} d{d} groups {ng} shared {shared}: std kernel {std_k.tolist()} gram {std_g.tolist()} rel {((std_g - std_k.to(std_g.device)).abs() / std_k.to(std_g.device)).max().item():.2e}; "
          f"out max diff {(a.float() - b.float()).abs().max().item():.3e} (|out| max {a.float().abs().max().item():.2f})", flush=True)

# ---- the HIP epilogue (dsc_linear_q_gram_f16) against the torch restatement and against linear_ln's q
import math
print("HIP epilogue:")
for (Bq, Bt, H, L, d, ng, K) in [(2, 2, 8, 4096, 40, 1, 320), (2, 2, 8, 1024, 80, 1, 640), (1, 2, 8, 4096, 40, 1, 320), (4, 4, 8, 1024, 40, 2, 320), (16, 16, 8, 1024, 80, 8, 640)]:
    S, C = 77, H * d
    x = torch.randn(Bq, L, K, generator=g).half().to(dev)
    w = (torch.randn(C, K, generator=g) / math.sqrt(K)).half().to(dev)
    b = (torch.randn(C, generator=g) * 0.1).half().to(dev)
    k = (torch.randn(Bt, S, C, generator=g) * 2.0 + 0.3).half().to(dev)
    gram = ops.xattn_gram_pack(k.view(Bt, S, H, d))
    q_ref = ops.linear(x, w, b, prefer_kernel=True)
    q, parts = ops.linear_q_gram(x, w, b, gram, H, Bt, ng)
    ref = ra.score_moment_partials(q_ref.view(Bq, L, H, d), k.view(Bt, S, H, d), ng)
    # the kernel's slots are per (row tile, column tile); the restatement's per row tile: compare the sums per (group, rep, row tile)
    nbq = C // 160
    got = parts.view(ng, -1, nbq, 2).sum(dim=2)
    rel = ((got - ref).abs() / ref.abs().clamp_min(1e-9)).max().item()
    tot_rel = ((parts.sum(1) - ref.sum(1)).abs() / ref.sum(1).abs()).max().item()
    # with the folded LayerNorm
    src = torch.randn(Bq, L, K, generator=g).half().to(dev)
    wi = (torch.eye(K) + 0.01 * torch.randn(K, K, generator=g)).half().to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev)
    s_, part = ops.linear_ln(src, wi, None, ln_stats=True)
    w2, b2, cvec = ops.fold_layernorm(w, b, gamma, beta)
    q_ln_ref = ops.linear_ln(s_, w2, b2, ln=(part, cvec, 1e-5))
    q_ln, parts_ln = ops.linear_q_gram(s_, w2, b2, gram, H, Bt, ng, ln=(part, cvec, 1e-5))
    ref_ln = ra.score_moment_partials(q_ln_ref.view(Bq, L, H, d), k.view(Bt, S, H, d), ng)
    tot_rel_ln = ((parts_ln.sum(1) - ref_ln.sum(1)).abs() / ref_ln.sum(1).abs()).max().item()
    print(f"Bq{Bq} Bt{Bt} L{L} d{d} K{K} groups {ng}: q equal {torch.equal(q, q_ref)} / ln {torch.equal(q_ln, q_ln_ref)}; partial pairs max rel {rel:.2e}, totals rel {tot_rel:.2e}, "
          f"with LN {tot_rel_ln:.2e}; deterministic {torch.equal(parts, ops.linear_q_gram(x, w, b, gram, H, Bt, ng)[1])}", flush=True)
    t_plain = tm_graph(lambda: ops.linear_ln(s_, w2, b2, ln=(part, cvec, 1e-5)))
    t_gram = tm_graph(lambda: ops.linear_q_gram(s_, w2, b2, gram, H, Bt, ng, ln=(part, cvec, 1e-5)))
    print(f"    hot: to_q (LayerNorm folded) {t_plain:6.2f} us, with the Gram epilogue {t_gram:6.2f} us", flush=True)
