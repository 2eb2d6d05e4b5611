"""GPU parity of the drop-in processors against the outputs of the REFERENCE's own processors
(tests/golden/processors.npz, captured by tests/golden/make_golden.py), and size-independent properties of the
region cross-attention at the full bench shape (Bc=2, H=8, L=4096, S=77, d=40)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from inputs import attn_inputs, ip_inputs, mask_inputs, proc_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 6e-3          # fp16 projections + fp16 attention output vs the reference's fp32 CPU run


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from diffusionspatialcontrol_amd import ops
    from diffusionspatialcontrol_amd.modules import attention_modify as am
    return ops, am


class DuckAttn:
    """the attributes the processors touch, fp16 on the GPU"""

    def __init__(self, p, residual_connection=False, rescale=1.0, self_kv=False):
        C, H = p["C"], p["H"]
        self.heads, self.scale = H, (C // H) ** -0.5
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = residual_connection, rescale
        self.upcast_attention = self.upcast_softmax = False

        def lin(w, b=None):
            m = nn.Linear(w.shape[1], w.shape[0], bias=b is not None)
            m.weight.data = torch.from_numpy(w)
            if b is not None:
                m.bias.data = torch.from_numpy(b)
            return m.half().cuda()

        self.to_q = lin(p["wq"])
        self.to_k = lin(p["wk_self"] if self_kv else p["wk"])
        self.to_v = lin(p["wv_self"] if self_kv else p["wv"])
        self.to_out = nn.ModuleList([lin(p["wo"], p["bo"]), nn.Dropout(0.0)])


@pytest.mark.parametrize("pname", ["ip2", "ip1"])
def test_ip_adapter_processors_against_reference_goldens(mods, pname):
    """IPAdapterAttnProcessor2_0 / IPAdapterAttnProcessor vs the reference's own outputs"""
    ops, am = mods
    from oracle import region_attention as ra
    g = np.load(os.path.join(G, "ip_processors.npz"))
    p, q = proc_inputs(), ip_inputs()
    L = p["L"]
    cls = am.IPAdapterAttnProcessor2_0 if pname == "ip2" else am.IPAdapterAttnProcessor
    proc = cls(hidden_size=p["C"], cross_attention_dim=p["ctx"], num_tokens=q["num_tokens"], scale=list(q["scale"]))
    assert sorted(proc.state_dict().keys()) == ["to_k_ip.0.weight", "to_k_ip.1.weight", "to_v_ip.0.weight", "to_v_ip.1.weight"]
    proc.load_state_dict({f"to_{kv}_ip.{i}.weight": torch.from_numpy(q[f"w{kv}_ip{i}"]) for kv in "kv" for i in range(2)})
    proc = proc.half().cuda()
    hs, enc = torch.from_numpy(p["hidden"]).half().cuda(), torch.from_numpy(p["enc"]).half().cuda()
    ips = [torch.from_numpy(q["ip0"]).half().cuda(), torch.from_numpy(q["ip1"]).half().cuda()]
    wf = lambda w, sigma, qk: w * sigma * qk.std()       # noqa: E731
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": wf}
    attn = DuckAttn(p)

    def check(out, key, tol=TOL):
        err = np.abs(out.float().cpu().numpy() - g[f"{pname}/{key}"])
        assert err.max() < tol, (key, err.max())

    with torch.no_grad():
        check(proc(attn, hs, encoder_hidden_states=(enc, ips), region_prompt=rp), "cross_region")
        check(proc(attn, hs, encoder_hidden_states=(enc, ips)), "cross_noregion")
        check(proc(attn, hs, encoder_hidden_states=torch.cat([enc, ips[0]], dim=1), region_prompt=rp), "cross_region_cat")
        with pytest.raises(ValueError):
            proc(attn, hs, encoder_hidden_states=(enc, ips), region_prompt=rp, ip_adapter_masks=torch.ones(2, 8, 8).cuda())
        with pytest.raises(ValueError):
            proc(attn, hs, encoder_hidden_states=(enc, ips), region_prompt=rp, ip_adapter_masks=torch.ones(3, 1, 8, 8).cuda())
        # masks: ones = identity; a half-plane mask against the oracle's restatement (downsample parity-unpinned)
        ones = torch.ones(2, 1, 16, 16).cuda()
        check(proc(attn, hs, encoder_hidden_states=(enc, ips), region_prompt=rp, ip_adapter_masks=ones), "cross_region", 8e-3)
        half = torch.zeros(2, 1, 16, 16)
        half[0, :, :, :8] = 1.0
        half[1, :, 8:, :] = 1.0
        out = proc(attn, hs, encoder_hidden_states=(enc, ips), region_prompt=rp, ip_adapter_masks=half.cuda())

        class DuckIP:
            num_tokens, scale = q["num_tokens"], list(q["scale"])
            to_k_ip = [lambda x, i=i: torch.nn.functional.linear(x, torch.from_numpy(q[f"wk_ip{i}"])) for i in range(2)]
            to_v_ip = [lambda x, i=i: torch.nn.functional.linear(x, torch.from_numpy(q[f"wv_ip{i}"])) for i in range(2)]

        class CpuAttn:
            heads, scale = p["H"], (p["C"] // p["H"]) ** -0.5
            residual_connection, rescale_output_factor = False, 1.0
            to_q = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wq"])))
            to_k = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wk"])))
            to_v = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wv"])))
            to_out = [lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wo"]), torch.from_numpy(p["bo"])), lambda x: x]

        fn = ra.ip_adapter_attn_processor2_0 if pname == "ip2" else ra.ip_adapter_attn_processor
        rp_cpu = dict(rp, weight_func=ra.default_weight_func)
        ref = fn(DuckIP, CpuAttn, torch.from_numpy(p["hidden"]), (torch.from_numpy(p["enc"]), [torch.from_numpy(q["ip0"]), torch.from_numpy(q["ip1"])]),
                 rp_cpu, ip_adapter_masks=half)
        assert (out.float().cpu() - ref).abs().max().item() < 8e-3
        # a self-attention call (no encoder states) passes through the text-less path
        o_self = proc(DuckAttn(p, self_kv=True), hs, region_prompt=rp)
        g0 = np.load(os.path.join(G, "processors.npz"))
        assert np.abs(o_self.float().cpu().numpy() - g0["p2/self" if pname == "ip2" else "p1/self"]).max() < TOL


@pytest.mark.parametrize("pname", ["p2", "p1"])
def test_processors_against_reference_goldens(mods, pname):
    ops, am = mods
    g = np.load(os.path.join(G, "processors.npz"))
    p = proc_inputs()
    L = p["L"]
    proc = am.AttnProcessor2_0() if pname == "p2" else am.AttnProcessor()
    hs, enc = torch.from_numpy(p["hidden"]).half().cuda(), torch.from_numpy(p["enc"]).half().cuda()
    wf = lambda w, sigma, qk: w * sigma * qk.std()       # noqa: E731  (app.py:1004)
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": wf}
    attn = DuckAttn(p)

    def check(out, key):
        err = np.abs(out.float().cpu().numpy() - g[f"{pname}/{key}"])
        assert err.max() < TOL, (key, err.max())

    with torch.no_grad():
        check(proc(attn, hs, encoder_hidden_states=enc, region_prompt=rp), "cross_region")
        check(proc(attn, hs, encoder_hidden_states=enc), "cross_noregion")
        rp_nd = dict(rp, region_state=torch.FloatTensor(0))
        check(proc(attn, hs, encoder_hidden_states=enc, region_prompt=rp_nd), "cross_nondict")
        check(proc(DuckAttn(p, self_kv=True), hs, region_prompt=rp), "self")
        h = int(math.isqrt(L))
        hs4 = hs.transpose(1, 2).reshape(2, p["C"], h, h).contiguous()
        rp4 = dict(rp, region_state={p["C"]: torch.from_numpy(p["w"])})      # keyed by shape[1] == C for 4-D input (:427)
        check(proc(DuckAttn(p, True, 2.0), hs4, encoder_hidden_states=enc, region_prompt=rp4), "cross_region_4d_res")
        with pytest.raises(KeyError):
            proc(attn, hs, encoder_hidden_states=enc, region_prompt=dict(rp, region_state={L + 1: torch.from_numpy(p["w"])}))


def test_attention_masks_against_reference_goldens(mods):
    """additive / boolean attention masks on the region path (reference attention_modify.py:85-91,144,448-452) against outputs
    of the reference's own functions (tests/golden/attention_masks.npz): the float mask enters the statistics
    (dsc_region_xattn_std_masked), the bool mask only rewrites itself, 4-D masks raise as in the reference."""
    ops, am = mods
    g = np.load(os.path.join(G, "attention_masks.npz"))
    x = attn_inputs("L256_d160", Bc=2, H=8, L=256, S=77, d=160)
    q, k, v = (torch.from_numpy(x[n]).half().cuda() for n in ("q", "k", "v"))
    w = torch.from_numpy(x["w"])
    m = mask_inputs()
    wf = lambda w_, s_, qk: w_ * s_ * qk.std()                      # noqa: E731
    call = lambda mask: am.scaled_dot_product_attention_regionstate(q, k, v, attn_mask=mask, weight_func=wf,  # noqa: E731
                                                                    region_state=w, sigma=torch.tensor(3.25))
    rows = x["rows"]
    for name in ("ls", "s1"):
        out = call(torch.from_numpy(m[name]).half().cuda())
        err = np.abs(out[:, :, rows, :].float().cpu().numpy() - g["a1/" + name])
        # fp16-rounding emulation on, |scores + mask + bias| reaches 16..32 where one fp16 ulp is 1.56e-2: one ulp of a logit
        assert err.max() < 1.6e-2 and err.mean() < 1e-3, (name, err.max(), err.mean())
    # the statistics really include the mask: the std of the masked scores differs from the unmasked one
    s0 = ops.region_xattn_std(q, k).item()
    s1 = ops.region_xattn_std(q, k, mask=torch.from_numpy(m["ls"]).cuda()).item()
    a = (q.float() @ k.float().transpose(-2, -1)) / math.sqrt(160) + torch.from_numpy(m["ls"]).cuda()
    assert abs(s1 - a.std().item()) < 2e-3 * s1 and abs(s1 - s0) > 0.05
    mb = torch.from_numpy(m["bool"]).cuda()
    out_b = call(mb)
    assert bool(mb.all())                                           # :86-87 turned every element True
    assert torch.equal(out_b, call(None))
    assert np.abs(out_b[:, :, rows, :].float().cpu().numpy() - g["a1/bool"]).max() < 8e-3
    with pytest.raises(RuntimeError):
        call(torch.zeros(2, 8, 1, 77, device="cuda", dtype=torch.half))
    # processors: AttnProcessor adds the [B*H, 1, S] mask inside get_attention_scores (std over the masked scores);
    # AttnProcessor2_0 raises with a region table and runs plain masked attention without one
    p = proc_inputs()
    L, S, H = p["L"], p["S"], p["H"]
    hs, enc = torch.from_numpy(p["hidden"]).half().cuda(), torch.from_numpy(p["enc"]).half().cuda()
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": wf}
    mp = torch.from_numpy(mask_inputs(L=L, S=S, BH=2 * H)["bh1s"]).half().cuda()
    attn = DuckAttn(p)
    attn.prepare_attention_mask = lambda mask, *a_, **k_: mask     # the golden's duck-typed attn returns the mask as is
    with torch.no_grad():
        o1 = am.AttnProcessor()(attn, hs, encoder_hidden_states=enc, attention_mask=mp, region_prompt=rp)
        assert np.abs(o1.float().cpu().numpy() - g["p1/cross_region_mask"]).max() < TOL
        o2 = am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, attention_mask=mp)
        assert np.abs(o2.float().cpu().numpy() - g["p2/cross_noregion_mask"]).max() < TOL
        with pytest.raises(RuntimeError):
            am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, attention_mask=mp, region_prompt=rp)


def test_custom_weight_func_takes_the_generic_path(mods):
    """a caller-supplied weight_func that is NOT w*sigma*std: evaluated as the reference would, result added by the kernel"""
    ops, am = mods
    p = proc_inputs()
    L = p["L"]
    hs, enc = torch.from_numpy(p["hidden"]).half().cuda(), torch.from_numpy(p["enc"]).half().cuda()
    wf = lambda w, sigma, qk: w * sigma * qk.abs().max() * 0.1      # noqa: E731
    assert not am.weight_func_is_default(wf)
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": wf}
    attn = DuckAttn(p)
    with torch.no_grad():
        out = am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, region_prompt=rp).float().cpu()
    # fp32 restatement with the same callable
    hs32, enc32 = torch.from_numpy(p["hidden"]), torch.from_numpy(p["enc"])
    q = hs32 @ torch.from_numpy(p["wq"]).t()
    k, v = enc32 @ torch.from_numpy(p["wk"]).t(), enc32 @ torch.from_numpy(p["wv"]).t()
    H, d = p["H"], p["C"] // p["H"]
    q4, k4, v4 = (t.view(2, -1, H, d).transpose(1, 2) for t in (q, k, v))
    a = (q4 @ k4.transpose(-2, -1)) / math.sqrt(d)
    flat = a.reshape(-1, L, 77)
    cw = wf(torch.from_numpy(p["w"]), torch.tensor(2.5), flat)
    flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
    o = (torch.softmax(flat.reshape(2, H, L, 77), -1) @ v4).transpose(1, 2).reshape(2, L, -1)
    ref = o @ torch.from_numpy(p["wo"]).t() + torch.from_numpy(p["bo"])
    assert (out - ref).abs().max().item() < 1.2e-2      # the callable sees fp16 scores (max is 1 fp16 ulp off)


def test_region_attention_function_signature(mods):
    """`scaled_dot_product_attention_regionstate(query, key, value, ..., weight_func, region_state, sigma)` drop-in"""
    ops, am = mods
    g = np.load(os.path.join(G, "attention_core.npz"))
    x = attn_inputs("L256_d160", Bc=2, H=8, L=256, S=77, d=160)
    q, k, v = (torch.from_numpy(x[n]).half().cuda() for n in ("q", "k", "v"))
    out = am.scaled_dot_product_attention_regionstate(q, k, v, weight_func=lambda w, s, qk: w * s * qk.std(),
                                                      region_state=torch.from_numpy(x["w"]), sigma=torch.tensor(3.25))
    err = np.abs(out[:, :, x["rows"], :].float().cpu().numpy() - g["L256_d160/out_rows"])
    assert err.max() < 8e-3                              # fp16-rounding emulation on: |scores + bias| reaches ~10 here


# ---------------------------------------------------------------------------- properties at the full bench shape
@pytest.fixture(scope="module")
def full(mods):
    ops, _ = mods
    Bc, H, L, S, d = 2, 8, 4096, 77, 40
    g = torch.Generator().manual_seed(11)
    q = torch.randn(Bc, L, H, d, generator=g).half().cuda()
    k = torch.randn(Bc, S, H, d, generator=g).half().cuda()
    v = torch.randn(Bc, S, H, d, generator=g).half().cuda()
    w = torch.zeros(2, L, S)
    w[:, 500:2000, 2:4] = 0.5
    w[:, 1500:3500, 4:6] += 0.75
    return ops, q, k, v, w


def _run(ops, q, k, v, w, sigma=4.0, packed=True):
    if packed:
        pk = ops.xattn_kv_pack(k, v)
        comp = ops.compress_region_table(w.cuda()) if w is not None else None
        return ops.region_xattn_packed(q, pk, k.shape[1], comp, sigma, ref_fp16_rounding=False)
    return ops.region_xattn(q, k, v, None if w is None else w.cuda(), sigma, layout="blhd", ref_fp16_rounding=False)


@pytest.mark.parametrize("packed", [True, False])
def test_rows_are_convex_combinations(full, packed):
    """softmax rows sum to one: V = 1 gives 1; V = c per channel gives c (both kernels, full size)"""
    ops, q, k, v, w = full
    ones = torch.ones_like(v)
    out = _run(ops, q, k, ones, w, packed=packed).float()
    assert (out - 1.0).abs().max().item() < 2e-3
    ramp = (torch.arange(v.shape[-1], device="cuda").half() / 8).expand_as(v).contiguous()
    out = _run(ops, q, k, ramp, w, packed=packed).float()
    assert (out - ramp[0, 0, 0].float()).abs().max().item() < 4e-3


def test_linear_in_v_and_key_permutation_invariant(full):
    ops, q, k, v, w = full
    g = torch.Generator(device="cuda").manual_seed(5)
    v2 = torch.randn(v.shape, generator=g, device="cuda").half()
    o1, o2 = _run(ops, q, k, v, w).float(), _run(ops, q, k, v2, w).float()
    o12 = _run(ops, q, k, (v.float() + v2.float()).half(), w).float()
    assert (o12 - (o1 + o2)).abs().max().item() < 6e-3
    perm = torch.randperm(77, generator=torch.Generator().manual_seed(2))
    op = _run(ops, q, k[:, perm.cuda()].contiguous(), v[:, perm.cuda()].contiguous(), w[:, :, perm].contiguous()).float()
    assert (op - o1).abs().max().item() < 3e-3        # fp16 P: summation order inside the MFMA changes


def test_sigma_zero_and_zero_table_equal_plain_attention(full):
    ops, q, k, v, w = full
    plain = _run(ops, q, k, v, None).float()
    assert (_run(ops, q, k, v, w, sigma=0.0).float() - plain).abs().max().item() < 1e-3
    assert (_run(ops, q, k, v, torch.zeros_like(w), sigma=9.0).float() - plain).abs().max().item() < 1e-3
    assert torch.equal(_run(ops, q, k, v, w), _run(ops, q, k, v, w))            # idempotent / bit-reproducible
    # a bias that is constant along the keys of a row cancels in the softmax
    const = torch.full_like(w, 0.3)
    assert (_run(ops, q, k, v, const, sigma=2.0).float() - plain).abs().max().item() < 2e-3


@pytest.mark.parametrize("n_groups", [1, 2])
def test_long_prompt_region_attention(mods, n_groups):
    """S = 154 keys (two 77-token chunks of the A1111-style encoder): beyond the fused kernels' 96 keys the processors run
    the reference's op sequence as library kernels; checked against the oracle incl. per-image std groups"""
    ops, am = mods
    from oracle import region_attention as ra
    g = torch.Generator().manual_seed(154 + n_groups)
    Bc, H, L, S, d = 4, 4, 256, 154, 40
    q = torch.randn(Bc, L, H, d, generator=g).half()
    k = torch.randn(Bc, S, H, d, generator=g).half()
    v = torch.randn(Bc, S, H, d, generator=g).half()
    w = torch.zeros(Bc, L, S)
    w[:, :100, 3:9] = 0.5
    w[:, 150:, 80:90] = -0.3
    ref = ra.region_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), w, 3.0,
                              n_std_groups=n_groups).transpose(1, 2)
    out = am._region_attention(q.cuda(), k.cuda(), v.cuda(), w, torch.tensor(3.0), None, "blhd", n_groups)
    assert out.shape == (Bc, L, H, d)
    assert (out.float().cpu() - ref).abs().max().item() < 4e-3
    # the processor takes this path for a long text and for long text without a region table
    p = proc_inputs()
    attn = DuckAttn(p)
    hs = torch.from_numpy(p["hidden"]).half().cuda()
    enc = torch.randn(2, 154, p["ctx"], generator=g).half().cuda()
    wt = torch.zeros(2, p["L"], 154)
    wt[:, :30, 5:12] = 0.7
    rp = {"region_state": {p["L"]: wt}, "sigma": torch.tensor(2.0), "weight_func": lambda w_, s_, qk: w_ * s_ * qk.std()}

    class CpuAttn:
        heads, scale = p["H"], (p["C"] // p["H"]) ** -0.5
        residual_connection, rescale_output_factor = False, 1.0
        to_q = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wq"])))
        to_k = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wk"])))
        to_v = staticmethod(lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wv"])))
        to_out = [lambda x: torch.nn.functional.linear(x, torch.from_numpy(p["wo"]), torch.from_numpy(p["bo"])), lambda x: x]

    with torch.no_grad():
        o = am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, region_prompt=rp).float().cpu()
        o_plain = am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc).float().cpu()
    rp_cpu = dict(rp, weight_func=ra.default_weight_func)
    r = ra.attn_processor2_0(CpuAttn, torch.from_numpy(p["hidden"]), enc.float().cpu(), rp_cpu)
    r_plain = ra.attn_processor2_0(CpuAttn, torch.from_numpy(p["hidden"]), enc.float().cpu(), None)
    assert (o - r).abs().max().item() < TOL and (o_plain - r_plain).abs().max().item() < TOL
