"""GPU parity of the fused region cross-attention (libdsc_hip.so, called through the C ABI) against the
CPU oracle on seeded inputs and against the golden rows captured from the reference itself.

Tolerance (fp16 path): the kernel rounds where the reference's fp16 tensors round, so the remaining
differences are fp32 summation order inside the MFMA and exp/reciprocal rounding, each of which can flip an fp16
rounding of a biased score.  One flipped rounding at score magnitude M moves that score by ulp16(M) = 2^(floor(log2 M)-10)
(0.0156 at M in [16, 32), which sigma = 14.6 reaches) and the output by about that much.  So:
|out - oracle_fp16| <= max(3e-3, 2 * ulp16(max |score + bias|)) with the MEAN error <= 3e-4; against the fp32
oracle / the fp32 reference goldens (no rounding emulation on either side): <= 6e-3, mean <= 4e-4.
"""
import math
import os

import numpy as np
import pytest
import torch

from inputs import attn_inputs
from oracle import region_attention as ra

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ATOL16, ATOL32 = 3e-3, 6e-3


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from diffusionspatialcontrol_amd import ops as _ops
    return _ops


def dev(x):
    return torch.from_numpy(x).cuda()


def tol16(q, k, w, sigma, n_std_groups=1):
    """max(3e-3, 2 ulp16 of the largest biased score) - see the module docstring."""
    d = q.shape[-1]
    a = (q @ k.transpose(-2, -1)) / math.sqrt(d)
    Bc, H, L, S = a.shape
    sd = ra.group_std(a, n_std_groups).float()
    wr = torch.repeat_interleave(w, (Bc * H) // w.shape[0], dim=0).reshape(Bc, H, L, S)
    m = (a + wr * sigma * sd[torch.arange(Bc) % n_std_groups].reshape(Bc, 1, 1, 1)).abs().max().item()
    return max(ATOL16, 2.0 * 2.0 ** (math.floor(math.log2(max(m, 1.0))) - 10))


CASES = [("L64_d160", 64, 160, 0.7), ("L256_d160", 256, 160, 3.25), ("L1024_d80", 1024, 80, 9.5),
         ("L4096_d40", 4096, 40, 14.6146)]


@pytest.mark.parametrize("name,L,d,sigma", CASES)
def test_against_reference_goldens(ops, name, L, d, sigma):
    g = np.load(os.path.join(G, "attention_core.npz"))
    x = attn_inputs(name, Bc=2, H=8, L=L, S=77, d=d)
    q, k, v = (dev(x[n]).half() for n in ("q", "k", "v"))
    out = ops.region_xattn(q, k, v, dev(x["w"]), sigma, ref_fp16_rounding=False).float().cpu()
    rows = x["rows"]
    err = (out[:, :, rows, :].numpy() - g[name + "/out_rows"])
    assert np.abs(err).max() < ATOL32, np.abs(err).max()
    assert np.abs(err).mean() < 4e-4
    sd = ops.region_xattn_std(q, k, ref_fp16_rounding=False).cpu()
    assert abs(sd.item() - float(g[name + "/std"])) < 1e-5 * float(g[name + "/std"]) + 1e-6
    cs = g[name + "/checksum"]
    assert abs(out.double().sum().item() - cs[0]) < 2e-4 * out.numel() ** 0.5 * 10 + 1.0
    assert abs((out.double() ** 2).sum().item() - cs[1]) < 2e-3 * cs[1]


@pytest.mark.parametrize("name,L,d,sigma", CASES)
@pytest.mark.parametrize("ref16", [True, False])
def test_against_oracle(ops, name, L, d, sigma, ref16):
    x = attn_inputs(name, Bc=2, H=8, L=L, S=77, d=d)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    if ref16:       # the reference's sigma is a 0-dim fp16 tensor (model_k_diffusion.py:1027-1029,1100): the caller
        sigma = float(torch.tensor(sigma).half())   # hands the kernel that already-rounded value
    exp = ra.region_attention(q, k, v, w, sigma, fp16_rounding=ref16)
    out = ops.region_xattn(q.cuda().half(), k.cuda().half(), v.cuda().half(), w.cuda(), sigma,
                           ref_fp16_rounding=ref16).float().cpu()
    err = (out - exp).abs()
    assert err.max().item() < (tol16(q, k, w, sigma) if ref16 else ATOL32), err.max().item()
    assert err.mean().item() < (3e-4 if ref16 else 4e-4), err.mean().item()


SHAPES = [
    # Bc, H, L, S, d, Bw, groups
    (2, 8, 100, 77, 40, 2, 1),      # ragged tail tile
    (1, 4, 33, 50, 64, 1, 1),       # S not a multiple of 16, one extra row
    (2, 2, 32, 96, 8, 4, 1),        # maximum S, minimum d, one table row per (b, h)
    (4, 8, 256, 77, 80, 4, 2),      # two images micro-batched: rows {i, 2+i} form std group i
    (2, 10, 512, 77, 64, 2, 1),     # SDXL-shaped heads
    (2, 20, 128, 77, 64, 1, 2),     # Bw = 1 broadcast, one group per row
    (3, 5, 70, 1, 16, 3, 3),        # a single key
    (2, 8, 4096, 77, 40, 2, 1),
]


@pytest.mark.parametrize("Bc,H,L,S,d,Bw,ng", SHAPES)
def test_shapes_and_groups(ops, Bc, H, L, S, d, Bw, ng):
    x = attn_inputs(f"shape/{Bc}/{H}/{L}/{S}/{d}", Bc=Bc, H=H, L=L, S=S, d=d, Bw=Bw)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    exp = ra.region_attention(q, k, v, w, 2.0, n_std_groups=ng, fp16_rounding=True)
    out = ops.region_xattn(q.cuda().half(), k.cuda().half(), v.cuda().half(), w.cuda(), 2.0, n_std_groups=ng)
    err = (out.float().cpu() - exp).abs()
    assert err.max().item() < tol16(q, k, w, 2.0, ng), err.max().item()
    assert err.mean().item() < 3e-4, err.mean().item()
    a = ((q @ k.transpose(-2, -1)).half().float() / math.sqrt(d)).half().float()
    sd = ops.region_xattn_std(q.cuda().half(), k.cuda().half(), n_std_groups=ng).cpu()
    exp_sd = ra.group_std(a, ng).half().float()
    assert torch.all((sd - exp_sd).abs() <= 1e-3 * exp_sd + 1e-6), (sd, exp_sd)


def test_projection_layout_and_device_sigma(ops):
    """q/k/v as strided views of [Bc, L, H*d] projection outputs; out lands in the [Bc, L, H*d] layout."""
    Bc, H, L, S, d = 2, 8, 256, 77, 40
    x = attn_inputs("proj_layout", Bc=Bc, H=H, L=L, S=S, d=d)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    exp = ra.region_attention(q, k, v, w, 1.25, fp16_rounding=True)
    qp = q.transpose(1, 2).reshape(Bc, L, H * d).cuda().half().contiguous()
    kp = k.transpose(1, 2).reshape(Bc, S, H * d).cuda().half().contiguous()
    vp = v.transpose(1, 2).reshape(Bc, S, H * d).cuda().half().contiguous()
    sig = torch.tensor([1.25], dtype=torch.float32, device="cuda")
    out = ops.region_xattn(qp.view(Bc, L, H, d), kp.view(Bc, S, H, d), vp.view(Bc, S, H, d), w.cuda(), sig,
                           layout="blhd")
    assert out.shape == (Bc, L, H, d) and out.is_contiguous()
    got = out.float().cpu().transpose(1, 2)
    assert (got - exp).abs().max().item() < ATOL16


def test_no_region_is_plain_sdpa(ops):
    x = attn_inputs("noregion", Bc=2, H=8, L=200, S=77, d=80)
    q, k, v = (torch.from_numpy(x[n]) for n in ("q", "k", "v"))
    exp = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    out = ops.region_xattn(q.cuda().half(), k.cuda().half(), v.cuda().half(), None, ref_fp16_rounding=False)
    assert (out.float().cpu() - exp).abs().max().item() < ATOL16
    # a zero table takes the region path and must agree with it (quirk q2: state None -> dict of zeros)
    out0 = ops.region_xattn(q.cuda().half(), k.cuda().half(), v.cuda().half(), torch.zeros(2, 200, 77).cuda(), 5.0,
                            ref_fp16_rounding=False)
    assert (out0.float() - out.float()).abs().max().item() < 1e-3


def test_bias_is_final(ops):
    """A caller-evaluated weight_func result is added as is."""
    x = attn_inputs("final", Bc=2, H=4, L=96, S=77, d=40)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    bias = w * 0.37
    a = (q @ k.transpose(-2, -1)) / math.sqrt(40) + torch.repeat_interleave(bias, 4, dim=0).reshape(2, 4, 96, 77)
    exp = torch.softmax(a, -1) @ v
    out = ops.region_xattn(q.cuda().half(), k.cuda().half(), v.cuda().half(), bias.cuda(), 123.0,
                           bias_is_final=True, ref_fp16_rounding=False)
    assert (out.float().cpu() - exp).abs().max().item() < ATOL32


def test_bit_reproducible(ops):
    x = attn_inputs("repro", Bc=2, H=8, L=1024, S=77, d=80)
    q, k, v = (torch.from_numpy(x[n]).cuda().half() for n in ("q", "k", "v"))
    w = torch.from_numpy(x["w"]).cuda()
    a = ops.region_xattn(q, k, v, w, 3.0)
    for _ in range(3):
        assert torch.equal(a, ops.region_xattn(q, k, v, w, 3.0))


def test_std_couples_rows(ops):
    """Sample 0's output changes when only sample 1's input changes (global std) - and stops doing so with one
    std group per row."""
    x = attn_inputs("L64_d160", Bc=2, H=8, L=64, S=77, d=160)
    q, k, v = (torch.from_numpy(x[n]).cuda().half() for n in ("q", "k", "v"))
    w = torch.from_numpy(x["w"]).cuda()
    q2 = q.clone()
    q2[1] *= 3.0
    o1, o2 = ops.region_xattn(q, k, v, w, 0.7), ops.region_xattn(q2, k, v, w, 0.7)
    assert (o1[0].float() - o2[0].float()).abs().max().item() > 1e-2
    s1, s2 = ops.region_xattn(q, k, v, w, 0.7, n_std_groups=2), ops.region_xattn(q2, k, v, w, 0.7, n_std_groups=2)
    assert torch.equal(s1[0], s2[0])


PACKED_SHAPES = [(2, 8, 4096, 77, 40, 2, 1), (2, 8, 1024, 77, 80, 2, 1), (2, 8, 256, 77, 160, 2, 1), (2, 8, 64, 77, 160, 2, 1),
                 (4, 8, 256, 77, 80, 4, 2), (2, 10, 512, 77, 64, 2, 1), (2, 2, 32, 96, 8, 4, 1), (3, 5, 70, 1, 16, 3, 3),
                 (2, 8, 100, 50, 40, 1, 1), (16, 8, 1024, 77, 40, 2, 8)]


@pytest.mark.parametrize("Bc,H,L,S,d,Bw,ng", PACKED_SHAPES)
def test_packed_path_equals_generic_and_oracle(ops, Bc, H, L, S, d, Bw, ng):
    """pre-packed K/V + compressed region table: bit-identical to the generic kernel (same arithmetic, same order),
    hence within the same tolerance of the oracle"""
    x = attn_inputs(f"packed/{Bc}/{H}/{L}/{S}/{d}", Bc=Bc, H=H, L=L, S=S, d=d, Bw=Bw)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    qd, kd, vd = q.cuda().half(), k.cuda().half(), v.cuda().half()
    gen = ops.region_xattn(qd, kd, vd, w.cuda(), 2.0, n_std_groups=ng)
    packed = ops.xattn_kv_pack(kd, vd, layout="bhld")
    comp = ops.compress_region_table(w.cuda())
    assert comp is not None and comp[1].shape[0] <= 32
    ids, rows = comp
    assert torch.equal(rows[ids.long()].reshape(w.shape), w.cuda())          # lossless
    q_blhd = qd.transpose(1, 2)                                              # [Bc, L, H, d] strided view
    out = ops.region_xattn_packed(q_blhd, packed, S, (ids, rows), 2.0, n_std_groups=ng)
    assert torch.equal(out.transpose(1, 2), gen)
    # the rows in the kernel's own table shape ([NU, 100], what the pipeline uploads: a flat 16-byte copy into LDS): same bits
    for r16 in (True, False):
        a_ = ops.region_xattn_packed(q_blhd, packed, S, (ids, rows), 2.0, n_std_groups=ng, ref_fp16_rounding=r16)
        b_ = ops.region_xattn_packed(q_blhd, packed, S, (ids, ops.pad_region_rows(rows)), 2.0, n_std_groups=ng, ref_fp16_rounding=r16)
        assert torch.equal(a_, b_)
    exp = ra.region_attention(q, k, v, w, 2.0, n_std_groups=ng, fp16_rounding=True)
    err = (out.transpose(1, 2).float().cpu() - exp).abs()
    assert err.max().item() < tol16(q, k, w, 2.0, ng)
    # fp32-score (lean) variant of both kernels: same result up to fp32 rounding of the exp2 / normalisation order
    lean_p = ops.region_xattn_packed(q_blhd, packed, S, (ids, rows), 2.0, n_std_groups=ng, ref_fp16_rounding=False)
    lean_g = ops.region_xattn(qd, kd, vd, w.cuda(), 2.0, n_std_groups=ng, ref_fp16_rounding=False)
    assert (lean_p.transpose(1, 2).float() - lean_g.float()).abs().max().item() < 1e-3   # (w*sigma)*std*log2e groups differently
    exp32 = ra.region_attention(q, k, v, w, 2.0, n_std_groups=ng)
    assert (lean_g.float().cpu() - exp32).abs().max().item() < ATOL32
    # no region: plain cross-attention through the packed image
    out0 = ops.region_xattn_packed(q_blhd, packed, S, None, ref_fp16_rounding=False)
    gen0 = ops.region_xattn(qd, kd, vd, None, ref_fp16_rounding=False)
    assert torch.equal(out0.transpose(1, 2), gen0)


@pytest.mark.parametrize("S", [3, 4, 5, 28, 31, 32, 33, 36, 37, 60, 63, 64, 65, 68, 69, 92, 95, 96])
def test_key_count_edges_of_the_lane_masks(ops, S):
    """The key-validity masks of the fp32-score kernels are built from scalar compares (lanes 0-31 hold key c, lanes 32-63 key
    c + 4 of every element: xattn_shared.h key_keep_mask): key counts on both sides of every 32-key tile edge and of the +4 lane
    half offset, packed forward and statistics against the fp32 oracle, packed against the generic kernel."""
    Bc, H, L, d = 2, 2, 100, 40
    x = attn_inputs(f"edges/{S}", Bc=Bc, H=H, L=L, S=S, d=d, Bw=2)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    qd, kd, vd = q.cuda().half(), k.cuda().half(), v.cuda().half()
    packed = ops.xattn_kv_pack(kd, vd, layout="bhld")
    ids, rows = ops.compress_region_table(w.cuda())
    q_blhd = qd.transpose(1, 2)
    for bias in ((ids, ops.pad_region_rows(rows)), None):
        lean_p = ops.region_xattn_packed(q_blhd, packed, S, bias, 2.0, ref_fp16_rounding=False)
        lean_g = ops.region_xattn(qd, kd, vd, w.cuda() if bias else None, 2.0, ref_fp16_rounding=False)
        assert torch.isfinite(lean_p).all()
        assert (lean_p.transpose(1, 2).float() - lean_g.float()).abs().max().item() < 1e-3
        exp32 = ra.region_attention(q, k, v, w if bias else torch.zeros_like(w), 2.0)
        assert (lean_p.transpose(1, 2).float().cpu() - exp32).abs().max().item() < ATOL32


LONG_SHAPES = [(2, 8, 4096, 154, 40, 2, 1), (2, 8, 1024, 231, 80, 2, 1), (2, 8, 256, 154, 160, 2, 1), (2, 8, 64, 231, 160, 2, 1),
               (4, 8, 1024, 154, 40, 4, 2), (2, 10, 512, 308, 64, 2, 1), (2, 2, 70, 97, 8, 1, 1), (2, 4, 100, 384, 16, 2, 1)]


@pytest.mark.parametrize("Bc,H,L,S,d,Bw,ng", LONG_SHAPES)
def test_long_prompt_chunked_kernels(ops, Bc, H, L, S, d, Bw, ng):
    """Prompts of several 77-token chunks (S = 154, 231, 308 ...; > 96 keys): the prepared-operand kernels walk the keys in
    chunks of 96 with an online softmax; global std over ALL keys.  Against the fp32 oracle (no fp16-rounding emulation on this
    path), with and without a region table, bit-reproducible."""
    x = attn_inputs(f"long/{Bc}/{H}/{L}/{S}/{d}", Bc=Bc, H=H, L=L, S=S, d=d, Bw=Bw)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    w[:, :, 100:104] += torch.from_numpy(x["w"][:, :, 2:3]) * 0.5 if S > 104 else 0.0     # bias in the later chunks too
    qd, kd, vd = q.cuda().half(), k.cuda().half(), v.cuda().half()
    packed = ops.xattn_kv_pack(kd, vd, layout="bhld")
    comp = ops.compress_region_table(w.cuda())
    assert comp is not None
    q_blhd = qd.transpose(1, 2)
    out = ops.region_xattn_packed(q_blhd, packed, S, comp, 2.0, n_std_groups=ng, ref_fp16_rounding=False)
    exp = ra.region_attention(q, k, v, w, 2.0, n_std_groups=ng)
    err = (out.transpose(1, 2).float().cpu() - exp).abs()
    assert err.max().item() < ATOL32 and err.mean().item() < 4e-4, (err.max().item(), err.mean().item())
    assert torch.equal(out, ops.region_xattn_packed(q_blhd, packed, S, comp, 2.0, n_std_groups=ng, ref_fp16_rounding=False))
    out0 = ops.region_xattn_packed(q_blhd, packed, S, None, ref_fp16_rounding=False)
    ref0 = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(d), dim=-1) @ v
    assert (out0.transpose(1, 2).float().cpu() - ref0).abs().max().item() < ATOL32
    with pytest.raises(Exception):
        ops.region_xattn_packed(q_blhd, packed, S, comp, 2.0, n_std_groups=ng, ref_fp16_rounding=True)   # emulation: <= 96 keys only


def test_compress_region_table_refuses_dense_tables(ops):
    w = torch.randn(2, 64, 77).cuda()
    assert ops.compress_region_table(w) is None
