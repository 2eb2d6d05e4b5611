"""The oracle (CPU restatement) against golden vectors captured from the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import math
import os
import types

import numpy as np
import pytest
import torch

from inputs import FakeTokenizer, attn_inputs, ip_inputs, proc_inputs, region_state_inputs
from oracle import k_diffusion_ref as kd
from oracle import region_attention as ra
from oracle import region_encoder as re_

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


ATTN_CASES = [("L64_d160", 64, 160), ("L256_d160", 256, 160), ("L1024_d80", 1024, 80), ("L4096_d40", 4096, 40)]


@pytest.mark.parametrize("name,L,d", ATTN_CASES)
def test_region_attention_core(name, L, d):
    g = load("attention_core.npz")
    x = attn_inputs(name, Bc=2, H=8, L=L, S=77, d=d)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    o = ra.region_attention(q, k, v, w, float(g[name + "/sigma"]))
    a = (q @ k.transpose(-2, -1)) / math.sqrt(d)
    assert abs(ra.group_std(a).item() - float(g[name + "/std"])) < 2e-6
    rows = g[name + "/rows"]
    np.testing.assert_array_equal(rows, x["rows"])
    np.testing.assert_allclose(o[:, :, rows, :].numpy(), g[name + "/out_rows"], atol=1e-5, rtol=0)
    cs = g[name + "/checksum"]
    od = o.double()
    assert abs(od.sum().item() - cs[0]) < 1e-5 * max(1.0, abs(cs[0])) + 1e-2
    assert abs((od * od).sum().item() - cs[1]) < 1e-5 * cs[1]
    if name + "/out" in g.files:
        np.testing.assert_allclose(o.numpy(), g[name + "/out"], atol=1e-5, rtol=0)


def test_std_couples_rows_of_a_group():
    """sample 0's output moves when only sample 1's input moves (global std)."""
    g = load("attention_core.npz")
    x = attn_inputs("L64_d160", Bc=2, H=8, L=64, S=77, d=160)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    q2 = q.clone()
    q2[1] *= 3.0
    o2 = ra.region_attention(q2, k, v, w, 0.7)
    np.testing.assert_allclose(o2[0].numpy(), g["coupling/out_b0"], atol=1e-5, rtol=0)
    assert np.abs(g["coupling/out_b0"] - g["L64_d160/out"][0]).max() > 1e-2
    # with one std group per row the coupling disappears
    o_sep = ra.region_attention(q, k, v, w, 0.7, n_std_groups=2)
    o2_sep = ra.region_attention(q2, k, v, w, 0.7, n_std_groups=2)
    np.testing.assert_allclose(o_sep[0].numpy(), o2_sep[0].numpy(), atol=1e-6)


def test_fp16_rounding_mode_tracks_reference_fp16():
    """The oracle's fp16-rounding mode against the reference run with fp16 tensors on CPU."""
    g = load("attention_core.npz")
    x = attn_inputs("L256_d160", Bc=2, H=8, L=256, S=77, d=160)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    o = ra.region_attention(q, k, v, w, 3.25, fp16_rounding=True)
    ref = g["fp16_L256_d160/out_rows"]
    got = o[:, :, x["rows"], :].numpy()
    # fp16 ulp at |x|~1 is 9.8e-4; the CPU half matmul may accumulate in a different order
    assert np.abs(got - ref).max() < 4e-3
    assert np.abs(got - ref).mean() < 2e-4
    a16 = ((q @ k.transpose(-2, -1)).half().float() / math.sqrt(160)).half().float()
    assert abs(ra.group_std(a16).half().item() - float(g["fp16_L256_d160/std"])) <= 1e-3


class DuckAttn:
    def __init__(self, p, residual_connection=False, rescale=1.0):
        self.heads = p["H"]
        self.scale = (p["C"] // p["H"]) ** -0.5
        self.residual_connection = residual_connection
        self.rescale_output_factor = rescale
        lin = lambda w, b=None: (lambda x: torch.nn.functional.linear(x, w, b))  # noqa: E731
        self.to_q = lin(torch.from_numpy(p["wq"]))
        self.to_k = lin(torch.from_numpy(p["wk"]))
        self.to_v = lin(torch.from_numpy(p["wv"]))
        self.to_out = [lin(torch.from_numpy(p["wo"]), torch.from_numpy(p["bo"])), lambda x: x]


@pytest.mark.parametrize("pname,fn", [("p2", ra.attn_processor2_0), ("p1", ra.attn_processor)])
def test_processors(pname, fn):
    g = load("processors.npz")
    p = proc_inputs()
    L = p["L"]
    hs, enc = torch.from_numpy(p["hidden"]), torch.from_numpy(p["enc"])
    sigma = torch.tensor(2.5)
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": sigma, "weight_func": ra.default_weight_func}
    attn = DuckAttn(p)
    tol = dict(atol=2e-5, rtol=0)
    np.testing.assert_allclose(fn(attn, hs, enc, rp).numpy(), g[pname + "/cross_region"], **tol)
    np.testing.assert_allclose(fn(attn, hs, enc, None).numpy(), g[pname + "/cross_noregion"], **tol)
    rp_nd = dict(rp, region_state=torch.FloatTensor(0))
    np.testing.assert_allclose(fn(attn, hs, enc, rp_nd).numpy(), g[pname + "/cross_nondict"], **tol)
    ps = dict(p, wk=p["wk_self"], wv=p["wv_self"])
    np.testing.assert_allclose(fn(DuckAttn(ps), hs, None, rp).numpy(), g[pname + "/self"], **tol)
    h = int(math.isqrt(L))
    hs4 = hs.transpose(1, 2).reshape(2, p["C"], h, h).contiguous()
    rp4 = dict(rp, region_state={p["C"]: torch.from_numpy(p["w"])})
    o = fn(DuckAttn(p, True, 2.0), hs4, enc, rp4)
    np.testing.assert_allclose(o.numpy(), g[pname + "/cross_region_4d_res"], **tol)
    # the table is looked up by hidden_states.shape[1]: a missing key is a KeyError (attention_modify.py:481)
    with pytest.raises(KeyError):
        fn(attn, hs, enc, dict(rp, region_state={L + 1: torch.from_numpy(p["w"])}))


class DuckIP:
    """num_tokens / scale / to_k_ip / to_v_ip of the reference's IP-Adapter processors (attention_modify.py:224-246)"""

    def __init__(self, q):
        self.num_tokens, self.scale = q["num_tokens"], list(q["scale"])
        lin = lambda w: (lambda x: torch.nn.functional.linear(x, torch.from_numpy(w)))  # noqa: E731
        self.to_k_ip = [lin(q[f"wk_ip{i}"]) for i in range(len(self.num_tokens))]
        self.to_v_ip = [lin(q[f"wv_ip{i}"]) for i in range(len(self.num_tokens))]


@pytest.mark.parametrize("pname,fn", [("ip2", ra.ip_adapter_attn_processor2_0), ("ip1", ra.ip_adapter_attn_processor)])
def test_ip_adapter_processors(pname, fn):
    """the oracle's IP-Adapter restatement against outputs of the reference's two IP-Adapter processors"""
    g = load("ip_processors.npz")
    p, q = proc_inputs(), ip_inputs()
    L = p["L"]
    hs, enc = torch.from_numpy(p["hidden"]), torch.from_numpy(p["enc"])
    ips = [torch.from_numpy(q["ip0"]), torch.from_numpy(q["ip1"])]
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": ra.default_weight_func}
    attn, ip = DuckAttn(p), DuckIP(q)
    tol = dict(atol=3e-5, rtol=0)
    np.testing.assert_allclose(fn(ip, attn, hs, (enc, ips), rp).numpy(), g[pname + "/cross_region"], **tol)
    np.testing.assert_allclose(fn(ip, attn, hs, (enc, ips), None).numpy(), g[pname + "/cross_noregion"], **tol)
    cat = torch.cat([enc, ips[0]], dim=1)
    np.testing.assert_allclose(fn(ip, attn, hs, cat, rp).numpy(), g[pname + "/cross_region_cat"], **tol)
    # the image branches matter: dropping them changes the result
    base = ra.attn_processor2_0(attn, hs, enc, rp).numpy()
    assert np.abs(base - g["ip2/cross_region"]).max() > 1e-2
    # mask validation errors of the reference (:637-647)
    with pytest.raises(ValueError):
        fn(ip, attn, hs, (enc, ips), rp, ip_adapter_masks=torch.ones(2, 8, 8))
    with pytest.raises(ValueError):
        fn(ip, attn, hs, (enc, ips), rp, ip_adapter_masks=torch.ones(3, 1, 8, 8))
    # all-ones masks are the identity, zero masks remove the image branch (the downsample itself is parity-unpinned)
    o1 = fn(ip, attn, hs, (enc, ips), rp, ip_adapter_masks=torch.ones(2, 1, 16, 16)).numpy()
    np.testing.assert_allclose(o1, g[pname + "/cross_region"], atol=1e-4, rtol=0)
    o0 = fn(ip, attn, hs, (enc, ips), rp, ip_adapter_masks=torch.zeros(2, 1, 16, 16)).numpy()
    ref0 = (ra.attn_processor2_0 if pname == "ip2" else ra.attn_processor)(attn, hs, enc, rp).numpy()
    np.testing.assert_allclose(o0, ref0, atol=3e-5, rtol=0)


def test_attention_masks():
    """the oracle's mask branches against outputs of the reference's own functions (tests/golden/make_golden_masks.py)"""
    from inputs import mask_inputs
    g = load("attention_masks.npz")
    x = attn_inputs("L256_d160", Bc=2, H=8, L=256, S=77, d=160)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    m = mask_inputs()
    rows = x["rows"]
    for name in ("ls", "s1"):
        o = ra.region_attention(q, k, v, w, 3.25, attn_mask=torch.from_numpy(m[name]))
        np.testing.assert_allclose(o[:, :, rows, :].numpy(), g["a1/" + name], atol=1e-5, rtol=0)
    # the bool branch never reaches the scores (:86-87 only rewrites the mask) and a 4-D float mask cannot be added in place
    assert bool(g["a1/bool_equals_unmasked"]) and bool(g["a1/bool_mask_all_true_afterwards"]) and bool(g["a1/b4_raises"])
    np.testing.assert_allclose(ra.region_attention(q, k, v, w, 3.25)[:, :, rows, :].numpy(), g["a1/bool"], atol=1e-5, rtol=0)
    p = proc_inputs()
    L, S, H = p["L"], p["S"], p["H"]
    hs, enc = torch.from_numpy(p["hidden"]), torch.from_numpy(p["enc"])
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": ra.default_weight_func}
    mp = torch.from_numpy(mask_inputs(L=L, S=S, BH=2 * H)["bh1s"])
    attn = DuckAttn(p)
    np.testing.assert_allclose(ra.attn_processor(attn, hs, enc, rp, attention_mask=mp).numpy(), g["p1/cross_region_mask"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(ra.attn_processor2_0(attn, hs, enc, None, attention_mask=mp).numpy(), g["p2/cross_noregion_mask"],
                               atol=2e-5, rtol=0)
    assert bool(g["p2/raises"])
    with pytest.raises(RuntimeError):
        ra.attn_processor2_0(attn, hs, enc, rp, attention_mask=mp)


def test_processor_variants_agree():
    """a3 and a4 are the same function of their inputs (max diff 0.0 in fp32)."""
    g = load("processors.npz")
    for k in ("cross_region", "cross_noregion", "self", "cross_region_4d_res"):
        np.testing.assert_allclose(g["p2/" + k], g["p1/" + k], atol=2e-6, rtol=0)


@pytest.mark.parametrize("name", list(region_state_inputs().keys()))
def test_region_encoder(name, capsys):
    g = load("region_encoder.npz")
    state, ids, W, H, nimg = region_state_inputs()[name]
    rs = re_.encode_region_map(FakeTokenizer(), 4, 8, True, state, W, H, nimg, text_ids=ids)
    if name + "/nondict_numel" in g.files:
        assert not isinstance(rs, dict) and rs.numel() == int(g[name + "/nondict_numel"])
        return
    assert sorted(rs.keys()) == g[name + "/keys"].tolist()
    for L, t in rs.items():
        assert t.dtype == torch.float32
        assert list(t.shape) == g[f"{name}/{L}/shape"].tolist()
        dense = np.zeros(t.shape, dtype=np.float32)
        idx = g[f"{name}/{L}/idx"]
        dense[idx[0], idx[1], idx[2]] = g[f"{name}/{L}/val"]
        np.testing.assert_array_equal(t.numpy(), dense)
        # quirk q1: uncond rows carry the same table as cond rows
        np.testing.assert_array_equal(t.numpy()[0::2], t.numpy()[1::2])


def test_region_encoder_quirks():
    cases = region_state_inputs()
    tok = FakeTokenizer()
    st, ids, W, H, n = cases["empty_region"]
    rs = re_.encode_region_map(tok, 4, 8, True, st, W, H, n, text_ids=ids)
    for L, t in rs.items():          # q3: `== max` with max == 0 is all-true -> whole level gets +weight
        col = t[1].abs().sum(0).nonzero().flatten()
        assert len(col) == 2 and torch.all(t[1][:, col] == 0.5)
    st, ids, W, H, n = cases["state_none"]
    rs = re_.encode_region_map(tok, 4, 8, True, st, W, H, n, text_ids=ids)
    assert isinstance(rs, dict) and all(float(t.abs().sum()) == 0 for t in rs.values())   # q2
    st, ids, W, H, n = cases["r4_512_n2"]
    rs = re_.encode_region_map(tok, 4, 8, True, st, W, H, n, text_ids=ids)
    assert rs[4096].shape == (4, 4096, 77)                                     # q4: [u,c,u,c]


def test_resize_block_aligned_is_kernel_independent():
    """For masks constant on aligned blocks the bicubic result is the block value."""
    rng = np.random.default_rng(3)
    for r in (8, 16, 32, 64):
        blocks = (rng.random((512 // 64, 512 // 64)) < 0.5).astype(np.uint8)
        m = np.kron(blocks, np.ones((64, 64), dtype=np.uint8))
        out = re_.resize_cubic_u8(m, (512 // r, 512 // r))
        exp = np.kron(blocks, np.ones((64 // r, 64 // r), dtype=np.uint8))
        np.testing.assert_array_equal(out, exp)


def test_resize_cubic_against_an_independent_implementation():
    """OpenCV is absent, so `resize_cubic_u8` (Keys kernel a = -0.75, half-pixel centres, replicated border, no antialias) stays
    parity unpinned against cv2 itself; torch's bicubic interpolate is an INDEPENDENT implementation of the same published kernel
    and conventions, and must give the same pixels up to the last rounding: arbitrary (non-aligned) sizes, down- and upscaling.
    (What this cannot see: OpenCV's uint8 path rounds the four coefficients to 11-bit fixed point; for the sizes the pipeline
    produces from multiples of 64 px every sampling offset is x.5 and those coefficients are exact.)"""
    rng = np.random.default_rng(11)
    for (H, W, h_r, w_r) in ((512, 512, 64, 64), (516, 500, 65, 63), (96, 80, 37, 52), (40, 56, 96, 120), (768, 512, 96, 64)):
        img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        got = re_.resize_cubic_u8(img, (w_r, h_r)).astype(np.int64)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).double()[None, None], size=(h_r, w_r), mode="bicubic", align_corners=False)[0, 0]
        ref = torch.clamp(torch.floor(ref + 0.5), 0, 255).numpy().astype(np.int64)
        diff = np.abs(got - ref)
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (H, W, h_r, w_r, diff.max(), (diff != 0).mean())
    # binary masks (what the encoder resizes): identical after the encoder's own binarisation at the maximum
    m = (rng.random((516, 500)) < 0.5).astype(np.uint8)
    m = np.kron(m[::12, ::10][:43, :50], np.ones((12, 10), dtype=np.uint8))
    got = re_.resize_cubic_u8(m, (63, 65))
    ref = torch.nn.functional.interpolate(torch.from_numpy(m).double()[None, None], size=(65, 63), mode="bicubic", align_corners=False)[0, 0]
    ref = torch.clamp(torch.floor(ref + 0.5), 0, 255).numpy().astype(np.uint8)
    assert (got != ref).mean() < 2e-3


def test_denoiser():
    g = load("denoiser.npz")
    den = kd.DiscreteEpsDenoiser(kd.sd15_alphas_cumprod())
    np.testing.assert_allclose(kd.sd15_alphas_cumprod().numpy(), g["alphas_cumprod"], rtol=1e-6)
    np.testing.assert_allclose(den.sigmas.numpy(), g["sigmas"], rtol=1e-6)
    assert abs(den.sigmas[0].item() - 0.029167533) < 1e-7 and abs(den.sigmas[-1].item() - 14.614646912) < 1e-5
    grid = torch.from_numpy(g["grid"])
    t = torch.stack([den.sigma_to_t(s.reshape(1)) for s in grid]).reshape(-1)
    np.testing.assert_allclose(t.numpy(), g["t_of_sigma"], atol=1e-3)
    tq = torch.stack([den.sigma_to_t(s.reshape(1), quantize=True) for s in grid]).reshape(-1)
    np.testing.assert_array_equal(tq.numpy(), g["t_of_sigma_quant"])
    c_out, c_in = den.get_scalings(grid)
    np.testing.assert_allclose(c_out.numpy(), g["c_out"], rtol=1e-6)
    np.testing.assert_allclose(c_in.numpy(), g["c_in"], rtol=1e-6)
    np.testing.assert_allclose(den.t_to_sigma(torch.from_numpy(g["t_grid"])).numpy(), g["sigma_of_t"], rtol=1e-5)
    np.testing.assert_allclose(den.get_sigmas(10).numpy(), g["get_sigmas_10"], rtol=1e-5)
    seen = {}

    def eps_fn(x, t, **kw):
        seen["x"], seen["t"] = x.clone(), t.clone()
        return torch.sin(x * 1.3) * 0.5 + 0.01 * t.reshape(-1, 1, 1, 1) / 1000.0

    out = den.forward(eps_fn, torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_sigma"]))
    np.testing.assert_allclose(out.numpy(), g["fwd_out"], atol=1e-5)
    np.testing.assert_allclose(seen["x"].numpy(), g["fwd_inner_x"], atol=1e-6)
    np.testing.assert_allclose(seen["t"].numpy(), g["fwd_inner_t"], atol=1e-3)
    assert int(g["b2_raises"]) == 1      # the reference's k-diffusion path cannot batch latents


def test_karras_known_answers():
    """SURVEY.md Appendix C (values computed in fp32 from the recalled formula; parity unpinned)."""
    s = kd.get_sigmas_karras(25, 0.029167533, 14.614646912)
    exp = [14.6146, 12.2830, 10.2778, 8.5600, 7.0944, 5.8494, 4.7965, 3.9105, 3.1686, 2.5508, 2.0392, 1.6183, 1.2741,
           0.9947, 0.7695, 0.5895, 0.4469, 0.3350, 0.2480, 0.1811, 0.1303, 0.0923, 0.0642, 0.0437, 0.0292, 0.0]
    np.testing.assert_allclose(s.numpy(), np.array(exp, dtype=np.float32), atol=6e-5)
    den = kd.DiscreteEpsDenoiser(kd.sd15_alphas_cumprod())
    assert abs(den.sigma_to_t(torch.tensor([14.6146])).item() - 998.9995) < 2e-2
    assert abs(den.sigma_to_t(torch.tensor([1.0])).item() - 353.8903) < 1e-2
    assert abs(math.sqrt(float(s[0]) ** 2 + 1) - 14.648815) < 1e-4


def test_dpmpp_2m_is_second_order():
    """x(sigma) = x0 + sigma*n has denoised == x0 exactly; a curved analytic denoiser converges with order 2."""
    x0 = torch.tensor([[0.3, -1.2]], dtype=torch.float64)

    def run(n):
        sig = kd.get_sigmas_karras(n, 0.05, 10.0).double()
        x = x0 + sig[0] * torch.tensor([[1.0, -0.5]], dtype=torch.float64)
        # denoiser of a Gaussian data distribution N(mu, s2): D = (s2*x + sigma^2*mu)/(s2 + sigma^2)
        mu, s2 = torch.tensor([[0.5, 0.25]], dtype=torch.float64), 0.6
        model = lambda x, s: (s2 * x + s[:, None] ** 2 * mu) / (s2 + s[:, None] ** 2)  # noqa: E731
        out = kd.sample_dpmpp_2m(model, x, sig[:-1].tolist() + [0.0])
        # exact probability-flow solution: (x - mu) scales with sqrt(s2 + sigma^2)
        s_end = float(sig[-2])
        exact_at_end = mu + (x - mu) * math.sqrt((s2 + s_end ** 2) / (s2 + float(sig[0]) ** 2))
        exact = (s2 * exact_at_end + s_end ** 2 * mu) / (s2 + s_end ** 2)      # final step returns the denoised
        return (out - exact).abs().max().item()

    e1, e2 = run(20), run(40)
    assert e2 < e1 / 3.0, (e1, e2)
