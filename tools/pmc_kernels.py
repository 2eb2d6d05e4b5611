"""PMC target for the SQ-counter pass of a round (round 4: every family of the step, not only attention + convolution):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
        SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_kernels.py

Launches, eagerly (one dispatch per row of the counter CSV), 20 times each, at the bench workload's shapes (Bc = 2):
  * region cross-attention: xp_stats<3> + xp_fwd<3> (L = 4096, d = 40, S = 77, row-table form the pipeline uploads)
  * flash self-attention L = 4096 d = 40 (K / V head-major)
  * conv3x3: 320 -> 320 @ 64x64 (320 workgroups) and 640 -> 640 @ 32x32 (160 workgroups)
  * gemm_tn_f16: M = 8192 N = 320 K = 320 (the most frequent linear of the step: to_out / to_q at 64x64),
    the three GEGLU projections (M = 8192 / 2048 / 512, K = 320 / 640 / 1280, N = 8K), the fused QKV projection at 64x64
`tools/make_mfma_busy.py` reduces the CSV to profiles/pmc_mfma_busy.json (bench.py's `mfma_busy_pct`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusionspatialcontrol_amd import ops

dev = "cuda"
N = 20
g = torch.Generator().manual_seed(3)
Bc, H, L, S, d = 2, 8, 4096, 77, 40
C = H * d
q = torch.randn(Bc, L, C, generator=g).half().to(dev)
k = torch.randn(Bc, S, C, generator=g).half().to(dev)
v = torch.randn(Bc, S, C, generator=g).half().to(dev)
w = torch.zeros(2, L, S)
w[:, 1000:2000, 2:4] = 0.5
w[:, 2500:3500, 4:6] = 0.5
sig = torch.tensor([7.0], device=dev)
out = torch.empty(Bc, L, H, d, dtype=torch.half, device=dev)
q4, k4, v4 = q.view(Bc, L, H, d), k.view(Bc, S, H, d), v.view(Bc, S, H, d)
packed = ops.xattn_kv_pack(k4, v4)
ids, rows = ops.compress_region_table(w, pad_rows=True)
comp = (ids.to(dev), ops.pad_region_rows(rows).to(dev))
for _ in range(N):
    ops.region_xattn_packed(q4, packed, S, comp, sig, n_std_groups=1, out=out, ref_fp16_rounding=False)
torch.cuda.synchronize()

qkv = torch.randn(Bc, L, 3 * C, generator=g).half().to(dev)
qs, ks, vs = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
ks, vs = (t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3) for t in (ks, vs))
qs = qs.contiguous()
for _ in range(N):
    ops.self_attention(qs, ks, vs)
torch.cuda.synchronize()

cl = torch.channels_last
for (cin, hw) in ((320, 64), (640, 32)):
    x = torch.randn(Bc, cin, hw, hw, generator=g).half().to(dev).contiguous(memory_format=cl)
    wt = (torch.randn(cin, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).half().to(dev).contiguous(memory_format=cl)
    for _ in range(N):
        ops.conv3x3(x, wt, None)
    torch.cuda.synchronize()

for (M, K, Nn, geglu) in ((8192, 320, 320, False), (8192, 320, 2560, True), (2048, 640, 5120, True), (512, 1280, 10240, True),
                          (8192, 320, 960, False)):
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    wt = (torch.randn(Nn, K, generator=g) / K ** 0.5).half().to(dev)
    b = torch.zeros(Nn).half().to(dev)
    for _ in range(N):
        ops.linear(x, wt, b, geglu=geglu, prefer_kernel=True)
    torch.cuda.synchronize()
print("pmc_kernels: done")
