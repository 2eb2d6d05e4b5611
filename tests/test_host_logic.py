"""CPU tests of the product's host-side logic (no GPU, no kernels): the modules/ mirror against the goldens
captured from the reference and against the oracle."""
import math
import os
import types

import numpy as np
import pytest
import torch

from inputs import FakeTokenizer, region_state_inputs
from diffusionspatialcontrol_amd import DscLibraryError
from diffusionspatialcontrol_amd.modules import attention_modify as am
from diffusionspatialcontrol_amd.modules import encode_region_map_function as er
from diffusionspatialcontrol_amd.modules import external_k_diffusion as ek
from diffusionspatialcontrol_amd.modules import sampling
from diffusionspatialcontrol_amd.modules.model_k_diffusion import ModelWrapper, SD15Scheduler, StableDiffusionPipeline
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import (Attention, UNet2DConditionLoadersMixin_modify,
                                                                         UNet2DConditionModel, UNetConfig)
from oracle import k_diffusion_ref as kd
from oracle import region_encoder as ore

G = os.path.join(os.path.dirname(__file__), "golden")


def _pipe_ns():
    return types.SimpleNamespace(tokenizer=FakeTokenizer(), unet=types.SimpleNamespace(down_blocks=[0, 1, 2, 3]),
                                 vae_scale_factor=8, do_classifier_free_guidance=True)


@pytest.mark.parametrize("name", list(region_state_inputs().keys()))
def test_region_encoder_matches_reference_goldens(name, capsys):
    g = np.load(os.path.join(G, "region_encoder.npz"))
    state, ids, W, H, nimg = region_state_inputs()[name]
    rs = er.encode_region_map(_pipe_ns(), state, width=W, height=H, num_images_per_prompt=nimg, text_ids=ids)
    if name + "/nondict_numel" in g.files:
        assert not isinstance(rs, dict) and rs.numel() == 0
        return
    assert sorted(rs.keys()) == g[name + "/keys"].tolist()
    for L, t in rs.items():
        assert t.dtype == torch.float32 and not t.is_cuda
        dense = np.zeros(g[f"{name}/{L}/shape"], dtype=np.float32)
        idx = g[f"{name}/{L}/idx"]
        dense[idx[0], idx[1], idx[2]] = g[f"{name}/{L}/val"]
        np.testing.assert_array_equal(t.numpy(), dense)
    if name == "map_none_notfound":
        assert "not found in text" in capsys.readouterr().out


def test_resize_matches_oracle_on_arbitrary_masks():
    """product (vectorised) vs oracle (loops) bicubic on non-aligned masks; cv2 itself is parity unpinned"""
    rng = np.random.default_rng(5)
    for (H, W, dsize) in [(512, 512, (64, 64)), (500, 700, (88, 63)), (96, 64, (12, 8)), (40, 40, (5, 5))]:
        yy, xx = np.mgrid[0:H, 0:W]
        m = (((xx - W * 0.4) ** 2 / (W * 0.3) ** 2 + (yy - H * 0.55) ** 2 / (H * 0.2) ** 2) < 1).astype(np.uint8)
        m |= (rng.random((H, W)) < 0.02).astype(np.uint8)
        np.testing.assert_array_equal(er._resize_cubic_u8(m, dsize), ore.resize_cubic_u8(m, dsize))


def test_denoiser_matches_reference_goldens():
    g = np.load(os.path.join(G, "denoiser.npz"))
    sched = SD15Scheduler()
    np.testing.assert_allclose(sched.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=1e-6)
    seen = {}

    class Inner:
        alphas_cumprod = sched.alphas_cumprod

        def apply_model(self, x, t, cond=None, **kw):
            seen["x"], seen["t"] = x.clone(), t.clone()
            return torch.sin(x * 1.3) * 0.5 + 0.01 * t.reshape(-1, 1, 1, 1) / 1000.0

    den = ek.CompVisDenoiser(Inner())
    np.testing.assert_allclose(den.sigmas.numpy(), g["sigmas"], rtol=1e-6)
    grid = torch.from_numpy(g["grid"])
    t = torch.stack([den.sigma_to_t(s.reshape(1)) for s in grid]).reshape(-1)
    np.testing.assert_allclose(t.numpy(), g["t_of_sigma"], atol=1e-3)
    np.testing.assert_allclose(den.sigma_to_t(grid).numpy(), g["t_of_sigma"], atol=1e-3)     # vectorised call
    tq = torch.stack([den.sigma_to_t(s.reshape(1), quantize=True) for s in grid]).reshape(-1)
    np.testing.assert_array_equal(tq.numpy(), g["t_of_sigma_quant"])
    for s, te in zip(g["grid"].tolist(), g["t_of_sigma"].tolist()):
        assert abs(den.sigma_to_t_host(s) - te) < 1e-3
        c_in, c_out, _ = den.step_scalars(s)
        assert abs(c_in - 1 / math.sqrt(s * s + 1)) < 1e-12 and c_out == -s
    c_out, c_in = den.get_scalings(grid)
    np.testing.assert_allclose(c_out.numpy(), g["c_out"], rtol=1e-6)
    np.testing.assert_allclose(c_in.numpy(), g["c_in"], rtol=1e-6)
    np.testing.assert_allclose(den.t_to_sigma(torch.from_numpy(g["t_grid"])).numpy(), g["sigma_of_t"], rtol=1e-5)
    np.testing.assert_allclose(den.get_sigmas(10).numpy(), g["get_sigmas_10"], rtol=1e-5)
    out = den(torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_sigma"]), cond=None)
    np.testing.assert_allclose(out.numpy(), g["fwd_out"], atol=1e-5)
    np.testing.assert_allclose(seen["x"].numpy(), g["fwd_inner_x"], atol=1e-6)
    np.testing.assert_allclose(seen["t"].numpy(), g["fwd_inner_t"], atol=1e-3)


def test_sampling_matches_oracle():
    s = sampling.get_sigmas_karras(25, 0.029167533, 14.614646912)
    np.testing.assert_allclose(s.numpy(), kd.get_sigmas_karras(25, 0.029167533, 14.614646912).numpy(), rtol=1e-6)
    sig = s.half().float().tolist()
    for (a, b, c), (a2, b2, c2) in zip(sampling.dpmpp_2m_coefficients(sig), kd.dpmpp_2m_coeffs(sig)):
        assert abs(a - a2) < 1e-12 and abs(b - b2) < 1e-12 and abs(c - c2) < 1e-12
    assert sampling.dpmpp_2m_coefficients(sig)[-1] == (0.0, 1.0, 0.0)
    assert sampling.append_dims(torch.ones(3), 4).shape == (3, 1, 1, 1)
    assert sampling.append_zero(torch.ones(2)).tolist() == [1.0, 1.0, 0.0]


def test_weight_func_probe():
    assert am.weight_func_is_default(lambda w, sigma, qk: w * sigma * qk.std())          # app.py:1004
    assert am.weight_func_is_default(lambda w, s, a: a.std() * s * w)
    assert not am.weight_func_is_default(lambda w, sigma, qk: w * sigma * qk.std(unbiased=False))
    assert not am.weight_func_is_default(lambda w, sigma, qk: w * sigma)
    assert not am.weight_func_is_default(lambda w, sigma, qk: w * sigma * qk.abs().max())
    assert not am.weight_func_is_default(lambda w, sigma, qk: 1 / 0)


def test_table_residency_cache_keys_on_identity_and_version():
    w = torch.zeros(2, 4, 3)
    a = am.resident_table(w, torch.device("cpu"))
    assert a is w                      # already fp32 contiguous on the target device: used in place
    w16 = torch.zeros(2, 4, 3, dtype=torch.float64)
    b1 = am.resident_table(w16, torch.device("cpu"))
    b2 = am.resident_table(w16, torch.device("cpu"))
    assert b1 is b2 and b1.dtype == torch.float32
    w16.add_(1.0)                      # in-place edit bumps _version -> fresh copy
    b3 = am.resident_table(w16, torch.device("cpu"))
    assert b3 is not b1 and float(b3.sum()) == 24.0


def test_unet_structure_and_processor_plumbing():
    with torch.device("meta"):
        sd15 = UNet2DConditionModel(UNetConfig.sd15())
    assert sum(p.numel() for p in sd15.parameters()) == 859_520_964          # published SD1.5 UNet size
    procs = sd15.attn_processors
    assert len(procs) == 32 and all(k.endswith(".processor") for k in procs)
    names = [k for k in procs if ".attn2." in k]
    assert len(names) == 16
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn2.processor" in procs
    assert "mid_block.attentions.0.transformer_blocks.0.attn1.processor" in procs
    assert isinstance(sd15, UNet2DConditionLoadersMixin_modify)
    assert sd15.down_blocks[0].attentions[0].transformer_blocks[0].attn2.to_k.weight.shape == (320, 768)
    assert len(sd15.down_blocks) == 4                                          # region-table levels (encode_region_map_sp)
    tiny = UNet2DConditionModel(UNetConfig.tiny())
    p = am.AttnProcessor()
    tiny.set_attn_processor(p)
    assert all(v is p for v in tiny.attn_processors.values())
    with pytest.raises(ValueError):
        tiny.set_attn_processor({"x.processor": p})
    d = {k: am.AttnProcessor2_0() for k in tiny.attn_processors}
    tiny.set_attn_processor(dict(d))
    assert all(tiny.attn_processors[k] is d[k] for k in d)
    # kwargs the processor does not declare are dropped (diffusers Attention.forward behaviour)
    seen = {}

    class P:
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, region_prompt=None):
            seen["rp"] = region_prompt
            return hidden_states

    a = Attention(32, None, 4, 8)
    a.set_processor(P())
    a(torch.zeros(1, 4, 32), region_prompt=1, something_else=2)
    assert seen["rp"] == 1


def test_product_path_fails_loudly_without_gpu():
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    with pytest.raises(DscLibraryError):
        unet(torch.zeros(2, 4, 16, 16, dtype=torch.float16), torch.tensor([10.0]), torch.zeros(2, 77, 64, dtype=torch.float16))
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    assert isinstance(pipe.k_diffusion_model, ek.CompVisDenoiser) and isinstance(pipe.k_diffusion_model.inner_model, ModelWrapper)
    assert pipe.get_scheduler("sample_dpmpp_2m") is sampling.sample_dpmpp_2m
    s = pipe.get_sigmas(25, {"scheduler": "karras"})
    assert s.shape == (26,) and float(s[-1]) == 0.0 and abs(float(s[0]) - 14.6146) < 1e-3
    with pytest.raises(NotImplementedError):
        pipe.txt2img("a prompt", num_inference_steps=2, sampler_name="sample_dpmpp_2m")
    emb = torch.zeros(1, 77, 64)
    with pytest.raises(DscLibraryError):
        pipe.txt2img(None, height=128, width=128, num_inference_steps=2, sampler_name="sample_dpmpp_2m",
                     sampler_opt={"scheduler": "karras"}, prompt_embeds=emb, negative_prompt_embeds=emb,
                     output_type="latent", fused=False)


def test_schedule_is_built_on_the_host_and_cached():
    """get_sigmas: sigma_min / sigma_max read from the denoiser once (a device read would synchronise every generation),
    the schedule computed on the host; _schedule: the model-dtype cast of it (reference model_k_diffusion.py:1027-1029)"""
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    kdm = pipe.k_diffusion_model
    for name in ("karras", "exponential", "polyexponential"):
        s = pipe.get_sigmas(12, {"scheduler": name})
        assert s.device.type == "cpu" and s.shape == (13,) and float(s[-1]) == 0.0
    assert pipe._sigma_range[0] is kdm and pipe._sigma_range[1] == kdm.sigmas[0].item() and pipe._sigma_range[2] == kdm.sigmas[-1].item()
    want = sampling.get_sigmas_karras(12, kdm.sigmas[0].item(), kdm.sigmas[-1].item())
    assert torch.equal(pipe.get_sigmas(12, {"scheduler": "karras"}), want)
    assert torch.equal(pipe.get_sigmas(12, {"scheduler": "karras", "discard_next_to_last_sigma": True}),
                       torch.cat([sampling.get_sigmas_karras(13, kdm.sigmas[0].item(), kdm.sigmas[-1].item())[:-2], want[-1:]]))
    d = pipe._schedule(12, {"scheduler": "karras"}, "cpu", torch.float16)
    assert d.dtype == torch.float16 and torch.equal(d, want.half())
    assert torch.equal(pipe.get_sigmas(12, {}), kdm.get_sigmas(12))          # the model's own discrete schedule otherwise
    pipe.k_diffusion_model = ek.CompVisDenoiser(pipe.k_diffusion_model.inner_model, quantize=False)    # a new denoiser: re-read
    pipe.get_sigmas(5, {"scheduler": "karras"})
    assert pipe._sigma_range[0] is pipe.k_diffusion_model


def test_vae_decoder_structure():
    """SD1.x AutoencoderKL decoder half: 49,490,199 parameters (decoder 49,490,179 + post_quant_conv 20), diffusers keys"""
    from diffusionspatialcontrol_amd.modules.vae_decoder import AutoencoderKLDecoder
    with torch.device("meta"):
        vae = AutoencoderKLDecoder()
    keys = set(vae.state_dict().keys())
    assert "decoder.mid_block.attentions.0.to_q.weight" in keys and "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in keys
    assert "decoder.up_blocks.0.upsamplers.0.conv.weight" in keys and "post_quant_conv.bias" in keys
    n_dec = sum(p.numel() for k, p in vae.state_dict().items() if k.startswith("decoder."))
    assert n_dec == 49_490_179, n_dec


def test_euler_scheduler_protocol_and_oracle_agree():
    """the diffusers-scheduler protocol object of modules/model_diffusers.py against oracle/diffusers_ref.py, and
    self-consistency of the Euler step (parity unpinned: diffusers is absent)"""
    import numpy as np
    from oracle import diffusers_ref
    from diffusionspatialcontrol_amd.modules.model_diffusers import EulerDiscreteScheduler
    for spacing in ("leading", "linspace", "trailing"):
        s = EulerDiscreteScheduler(timestep_spacing=spacing)
        s.set_timesteps(25)
        ts, sig, init = diffusers_ref.euler_schedule(25, spacing)
        np.testing.assert_array_equal(s.timesteps.numpy(), ts)
        np.testing.assert_array_equal(s.sigmas.numpy(), sig)
        assert abs(float(s.init_noise_sigma) - init) < 1e-6
        assert s.sigmas[-1] == 0 and bool((s.sigmas[:-1] > s.sigmas[1:]).all())
        assert s.sigmas.device.type == "cpu" and s.sigmas.dtype == torch.float32      # the sigma the processors receive (:352)
    s = EulerDiscreteScheduler(timestep_spacing="linspace")
    s.set_timesteps(10)
    assert abs(float(s.sigmas[0]) - 14.6146) < 1e-3                      # sigma_max of the SD1.x schedule
    s = EulerDiscreteScheduler()                                           # "leading" + offset 1: first timestep 901
    s.set_timesteps(10)
    assert float(s.timesteps[0]) == 901.0 and abs(float(s.init_noise_sigma) - (float(s.sigmas[0]) ** 2 + 1) ** 0.5) < 1e-5
    # Euler on the exact denoiser of a point mass at x0 (eps = (x - x0) / sigma) lands on x0 at sigma = 0
    x0 = torch.tensor([[0.3, -1.2, 2.0]])
    x = x0 + float(s.sigmas[0]) * torch.tensor([[1.0, -0.5, 0.25]])
    for i, t in enumerate(s.timesteps):
        sigma = float(s.sigmas[i])
        assert abs(float(s.scale_model_input(torch.ones(1), t)) - 1 / (sigma ** 2 + 1) ** 0.5) < 1e-6
        x = s.step((x - x0) / sigma, t, x)[0]
    assert torch.allclose(x, x0, atol=1e-5)


def test_prompt_parser_against_reference_goldens():
    """modules/prompt_parser.py vs outputs of the reference's own prompt_parser.py: emphasis
    parsing incl. its quirks, 75-token chunking with comma back-tracking and BREAK, weighted encoding with clip skip"""
    import json
    import numpy as np
    from inputs import FakeClipTokenizer, fake_text_encoder, prompt_cases
    from diffusionspatialcontrol_amd.modules import prompt_parser as pp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "prompt_parser.npz"))
    parsed = json.loads(bytes(g["parse_json"]).decode())
    assert len(parsed) >= 16
    for text, want in parsed.items():
        if want == "ValueError":
            with pytest.raises(ValueError):
                pp.parse_prompt_attention(text)
        else:
            assert pp.parse_prompt_attention(text) == want, text
    chunks = json.loads(bytes(g["chunk_json"]).decode())
    tok, enc = FakeClipTokenizer(), fake_text_encoder()
    emb = pp.FrozenCLIPEmbedderWithCustomWords(tok, enc, 1)
    assert emb.comma_token == 267 and emb.id_start == 49406 and emb.id_end == 49407
    for text, want in chunks.items():
        ch, n = emb.tokenize_line(text)
        assert n == want["count"], text
        assert [c.tokens for c in ch] == want["tokens"] and [c.multipliers for c in ch] == want["mult"], text
        assert all(len(c.tokens) == 77 for c in ch)
    for clip_skip in (1, 2):
        emb = pp.FrozenCLIPEmbedderWithCustomWords(tok, enc, clip_skip)
        for k, pair in enumerate(prompt_cases()["encode"]):
            with torch.no_grad():
                ids, z = emb(pair)
            np.testing.assert_array_equal(np.asarray(ids), g[f"encode/{clip_skip}/{k}/ids"])
            np.testing.assert_allclose(z.numpy(), g[f"encode/{clip_skip}/{k}/z"], rtol=1e-6, atol=1e-6)
    assert emb.get_target_prompt_token_count(0) == 75 and emb.get_target_prompt_token_count(76) == 150


# ----------------------------------------------------------------------------- samplers (callers of the hot path)
def _gauss_model(s0):
    """exact denoiser for N(0, s0^2) data; the probability-flow ODE then has the closed form
    x(sigma) = x(sigma_max) sqrt((s0^2 + sigma^2) / (s0^2 + sigma_max^2))"""
    return lambda x, sigma, **kw: x * (s0 ** 2 / (s0 ** 2 + sigma.reshape(-1, *([1] * (x.ndim - 1))).to(x.dtype) ** 2))


def _ode_error(fn, n, s0=1.3, **kw):
    sig = sampling.get_sigmas_karras(n, 0.03, 14.6).double()
    x = torch.full((2, 4), 1.0, dtype=torch.float64) * sig[0]
    out = fn(_gauss_model(s0), x, sig, **kw)
    return (out - x * math.sqrt(s0 ** 2 / (s0 ** 2 + float(sig[0]) ** 2))).abs().max().item()


@pytest.mark.parametrize("name,kw,min_order", [
    ("sample_euler", {}, 0.9), ("sample_heun", {}, 1.9), ("sample_dpm_2", {}, 1.9), ("sample_lms", {}, 2.5),
    ("sample_euler_ancestral", {"eta": 0.0}, 0.9), ("sample_dpm_2_ancestral", {"eta": 0.0}, 1.9),
    ("sample_dpmpp_2s_ancestral", {"eta": 0.0}, 1.5), ("sample_dpmpp_sde", {"eta": 0.0}, 1.5),
    ("sample_dpmpp_2m_sde", {"eta": 0.0}, 1.9), ("sample_dpmpp_2m_sde", {"eta": 0.0, "solver_type": "heun"}, 1.9),
    ("sample_dpmpp_3m_sde", {"eta": 0.0}, 1.9)])
def test_sampler_convergence_order(name, kw, min_order):
    """The k-diffusion samplers are un-vendored (parity unpinned): each restatement must at least solve the probability-flow
    ODE of an analytic denoiser at its published order (eta = 0 turns the ancestral / SDE samplers into ODE solvers)."""
    fn = getattr(sampling, name)
    e20, e40 = _ode_error(fn, 20, **kw), _ode_error(fn, 40, **kw)
    assert e40 < e20 and math.log2(e20 / e40) > min_order, (e20, e40)
    assert e40 < 6e-2


def test_sampler_eta0_reductions_and_schedules():
    sig = sampling.get_sigmas_karras(12, 0.03, 14.6).double()
    x = torch.randn(3, 5, generator=torch.Generator().manual_seed(1), dtype=torch.float64) * sig[0]
    m = _gauss_model(0.8)
    assert torch.allclose(sampling.sample_euler_ancestral(m, x, sig, eta=0.0), sampling.sample_euler(m, x, sig), atol=1e-12)
    assert torch.allclose(sampling.sample_dpm_2_ancestral(m, x, sig, eta=0.0), sampling.sample_dpm_2(m, x, sig), atol=1e-12)
    # DPM++ 2M SDE without noise is DPM++ 2M: same coefficients as the fused step's host scalars
    co = sampling.dpmpp_2m_coefficients([float(v) for v in sig])
    y, old = x.clone(), None
    for i, (a, b, c) in enumerate(co):
        d = m(y, sig[i] * y.new_ones(3))
        y = a * y + b * d + (c * old if old is not None else 0.0)
        old = d
    assert torch.allclose(sampling.sample_dpmpp_2m_sde(m, x, sig, eta=0.0), y, atol=1e-9)
    e = sampling.get_sigmas_exponential(10, 0.03, 14.6)
    pe = sampling.get_sigmas_polyexponential(10, 0.03, 14.6, rho=1.0)
    assert e.shape == (11,) and float(e[-1]) == 0.0 and torch.allclose(e, pe, rtol=1e-5)
    assert abs(float(e[0]) - 14.6) < 1e-4 and abs(float(e[-2]) - 0.03) < 1e-6
    assert torch.allclose(torch.log(e[:-1]).diff(), torch.log(e[:-1]).diff()[0].expand(9), atol=1e-5)
    assert sampling.get_ancestral_step(2.0, 1.0, eta=0.0) == (1.0, 0.0)
    dn, up = sampling.get_ancestral_step(2.0, 1.0, eta=1.0)
    assert abs(dn ** 2 + up ** 2 - 1.0) < 1e-12 and abs(up ** 2 - 0.75) < 1e-12
    assert abs(sum(sampling.linear_multistep_coeff(3, [4.0, 3.0, 2.5, 1.0], 2, j) for j in range(3)) - (1.0 - 2.5)) < 1e-12


@pytest.mark.parametrize("name,n", [("sample_euler_ancestral", 120), ("sample_dpm_2_ancestral", 30),
                                    ("sample_dpmpp_2s_ancestral", 30), ("sample_dpmpp_sde", 30), ("sample_dpmpp_2m_sde", 30),
                                    ("sample_dpmpp_3m_sde", 30)])
def test_stochastic_samplers_keep_the_marginal(name, n):
    """With the exact denoiser of N(0, s0^2) data every stochastic sampler must end with samples of standard deviation s0
    (first-order Euler a needs more steps for the same tolerance: 0.90 s0 at 30 steps, 0.97 at 120).  DPM++ SDE only passes
    with noise increments of ONE Brownian path (its two queries per step overlap): independent draws give 0.92."""
    s0 = 1.3
    sig = sampling.get_sigmas_karras(n, 0.03, 14.6).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(20000, 1, generator=g, dtype=torch.float64) * math.sqrt(s0 ** 2 + float(sig[0]) ** 2)
    torch.manual_seed(11)
    out = getattr(sampling, name)(_gauss_model(s0), x, sig)
    assert abs(out.std().item() / s0 - 1.0) < 0.04, out.std().item()
    assert abs(out.mean().item()) < 0.05


def test_extra_samplers_match_reference_goldens():
    """restart / DDPM / LCM / Heun++ against outputs of the REFERENCE's samplers_extra_k_diffusion.py
    (tests/golden/make_golden_samplers.py) on the analytic denoiser of tests/golden/inputs.py"""
    import numpy as np
    from inputs import analytic_denoiser, sampler_cases, sampler_start
    from diffusionspatialcontrol_amd.modules import samplers_extra_k_diffusion as sx
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers_extra.npz"))
    for name, (fn, n, kw, seed) in sampler_cases().items():
        x, sig = sampler_start(n)
        calls = []

        def model(xx, s, **k):
            calls.append(float(s.reshape(-1)[0]))
            return analytic_denoiser(xx, s)
        torch.manual_seed(seed)
        y = getattr(sx, fn)(model, x.clone(), sig, disable=True, **kw)
        assert len(calls) == len(gold[name + "/model_sigmas"]), name
        assert np.abs(np.array(calls) - gold[name + "/model_sigmas"]).max() < 1e-6, name
        assert np.abs(y.numpy() - gold[name]).max() < 1e-6, (name, np.abs(y.numpy() - gold[name]).max())


def test_pipeline_sampler_plumbing():
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    for sched in ("karras", "exponential", "polyexponential", None):
        s = pipe.get_sigmas(10, {"scheduler": sched} if sched else {})
        assert s.shape == (11,) and float(s[-1]) == 0.0 and bool((s[:-1].diff() < 0).all())
    s = pipe.get_sigmas(10, {"scheduler": "karras", "discard_next_to_last_sigma": True})
    full = pipe.get_sigmas(11, {"scheduler": "karras"})
    assert s.shape == (11,) and torch.equal(s[:-1], full[:-2]) and float(s[-1]) == 0.0
    for name in ("sample_euler", "sample_euler_ancestral", "sample_lms", "sample_heun", "sample_dpm_2",
                 "sample_dpm_2_ancestral", "sample_dpmpp_2s_ancestral", "sample_dpmpp_2m", "sample_dpmpp_sde",
                 "sample_dpmpp_2m_sde", "sample_dpmpp_3m_sde"):                       # every k-diffusion name app.py:170-220 lists
        assert callable(pipe.get_scheduler(name))
    x = torch.zeros(1, 4, 8, 8)
    sig = pipe.get_sigmas(5, {"scheduler": "karras"})
    ex = pipe.get_sampler_extra_args_t2i(sig, 0.3, 5, {"brownian_noise": True, "solver_type": "heun"}, x, 7,
                                         sampling.sample_dpmpp_2m_sde)
    assert ex["eta"] == 0.3 and ex["solver_type"] == "heun" and ex["sigmas"] is sig
    n1, n2 = ex["noise_sampler"](sig[0], sig[1]), ex["noise_sampler"](sig[1], sig[2])
    again = pipe.create_noise_sampler(x, sig, 5, 7)
    assert n1.shape == x.shape and not torch.equal(n1, n2) and torch.equal(again(sig[0], sig[1]), n1)   # seeded


def test_vae_structure_and_image_side_helpers():
    """AutoencoderKL: the published SD1.x parameter total, diffusers key names; the image-side helpers of img2img /
    inpainting (reference model_k_diffusion.py:458-481, 916-941, 1293-1362) on CPU tensors"""
    from diffusionspatialcontrol_amd.modules.vae_decoder import AutoencoderKL, DiagonalGaussianDistribution
    with torch.device("meta"):
        vae = AutoencoderKL()
    assert sum(p.numel() for p in vae.parameters()) == 83653863
    assert sum(p.numel() for p in vae.encoder.parameters()) == 34163592
    keys = set(vae.state_dict().keys())
    assert {"encoder.conv_in.weight", "encoder.down_blocks.2.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_q.weight",
            "encoder.conv_norm_out.weight", "encoder.conv_out.bias", "quant_conv.weight", "post_quant_conv.bias"} <= keys
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys
    d = DiagonalGaussianDistribution(torch.cat([torch.ones(1, 4, 2, 2), torch.full((1, 4, 2, 2), 50.0)], dim=1))
    assert float(d.logvar.max()) == 20.0 and torch.equal(d.mode(), torch.ones(1, 4, 2, 2))
    s1, s2 = d.sample(torch.Generator().manual_seed(1)), d.sample(torch.Generator().manual_seed(1))
    assert torch.equal(s1, s2)
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    sig = pipe.get_sigmas(6, {"scheduler": "karras"})
    ex = pipe.get_sampler_extra_args_i2i(sig[2:], 5, {}, torch.zeros(1, 4, 8, 8), 0, sampling.sample_dpmpp_2m)
    assert set(ex) == {"sigmas"} and ex["sigmas"].shape == (5,)
    a, st = pipe._sigma_to_alpha_sigma_t(3.0)
    assert abs(a - 10 ** -0.5) < 1e-12 and abs(st - 3 * 10 ** -0.5) < 1e-12
    lat = torch.ones(1, 4, 8, 8)
    out = pipe.prepare_latents_inpating(1, 4, 64, 64, torch.float32, "cpu", None, latents=lat, image=lat * 2,
                                        sigma=torch.tensor(2.0), is_strength_max=True, return_noise=True, return_image_latents=True)
    assert torch.allclose(out[0], lat * 5 ** 0.5) and torch.equal(out[1], lat) and torch.equal(out[2], lat * 2)
    g = torch.Generator().manual_seed(3)
    out = pipe.prepare_latents_inpating(1, 4, 64, 64, torch.float32, "cpu", g, image=lat * 2, sigma=torch.tensor(2.0),
                                        is_strength_max=False, return_noise=True, return_image_latents=True)
    assert torch.allclose(out[0], lat * 2 + 2.0 * out[1])                       # image + sigma * noise, no extra scaling
    with pytest.raises(ValueError):
        pipe.prepare_latents_inpating(1, 4, 64, 64, torch.float32, "cpu", None, is_strength_max=False)
    m = pipe._image_tensor(np.array([[0.2, 0.7], [0.5, 0.4]], dtype=np.float32), 4, 4, mask=True)
    assert m.shape == (1, 1, 4, 4) and m[0, 0, 0, 3] == 1.0 and m[0, 0, 0, 0] == 0.0 and m[0, 0, 3, 0] == 1.0
    im = pipe._image_tensor(np.zeros((4, 4, 3), dtype=np.float32), 4, 4)
    assert im.shape == (1, 3, 4, 4) and float(im.max()) == -1.0
    with pytest.raises(NotImplementedError):
        pipe._encode_vae_image(torch.zeros(1, 3, 8, 8), None)


def test_v_prediction_denoiser_matches_reference_golden():
    """DiscreteVDDPMDenoiser / CompVisVDenoiser (reference external_k_diffusion.py:142-182) against outputs of the reference's
    own classes (tests/golden/vdenoiser.npz), and the pipeline's `setup_unet` choice (:138-141)"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vdenoiser.npz"))
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    seen = []

    class Inner:
        alphas_cumprod = acp

        def apply_model(self, x, t, cond=None, **kw):
            seen.append((x.clone(), t.clone(), sorted(kw)))
            return (torch.cos(x * 0.9) * 0.4 - 0.02 * t.reshape(-1, 1, 1, 1) / 1000.0)[:, :4]

    den = ek.CompVisVDenoiser(Inner())
    assert np.allclose(den.sigmas.numpy(), g["sigmas"], rtol=1e-6)
    for name, v in zip(("c_skip", "c_out", "c_in"), den.get_scalings(torch.from_numpy(g["grid"]))):
        assert np.allclose(v.numpy(), g[name], rtol=1e-6, atol=1e-7), name
    out = den(torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_sigma"]), cond=None, cross_attention_kwargs={"k": 1})
    assert np.abs(out.numpy() - g["fwd_out"]).max() < 1e-5
    assert np.abs(seen[-1][0].numpy() - g["fwd_inner_x"]).max() < 1e-6 and np.abs(seen[-1][1].numpy() - g["fwd_inner_t"]).max() < 1e-3
    assert len(seen[-1][2]) == int(g["kwargs_reach_the_model"]) == 0          # the reference drops the kwargs: kept (documented)
    den.pass_kwargs = True
    den(torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_sigma"]), cond=None, cross_attention_kwargs={"k": 1})
    assert seen[-1][2] == ["cross_attention_kwargs"]
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler(prediction_type="v_prediction"))
    assert isinstance(pipe.k_diffusion_model, ek.CompVisVDenoiser) and pipe.v_prediction
    with pytest.raises(NotImplementedError):
        pipe._denoise_fused(None, None, None, None, None, 7.5, 1, {}, -1, 0)


def test_controlnet_structure_and_schedule():
    """ControlNetModel: the published SD1.5 ControlNet size, diffusers key names, zero-initialised output convolutions; the
    pipeline's keep schedule / conditioning scales (reference model_k_diffusion.py:355-427)"""
    from diffusionspatialcontrol_amd.modules.controlnet import ControlNetModel, MultiControlNetModel
    with torch.device("meta"):
        cn = ControlNetModel()
    assert sum(p.numel() for p in cn.parameters()) == 361279120
    keys = set(cn.state_dict().keys())
    assert {"controlnet_cond_embedding.conv_in.weight", "controlnet_cond_embedding.blocks.5.bias",
            "controlnet_cond_embedding.conv_out.weight", "controlnet_down_blocks.11.weight", "controlnet_mid_block.bias",
            "down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight", "mid_block.resnets.1.conv2.weight"} <= keys
    assert not any(k.startswith("up_blocks") or k.startswith("conv_out") for k in keys)
    tiny = ControlNetModel(UNetConfig.tiny())
    assert len(tiny.controlnet_down_blocks) == 12 and float(tiny.controlnet_mid_block.weight.detach().abs().max()) == 0.0
    assert float(tiny.controlnet_cond_embedding.conv_out.weight.detach().abs().max()) == 0.0
    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    pipe.setup_controlnet(tiny)
    img = torch.rand(1, 3, 32, 32)
    im, keep, guess, scale = pipe.preprocess_controlnet(0.8, 0.0, 0.5, img, 64, 64, 4, 1, 1)
    assert im.shape == (2, 3, 64, 64) and keep == [1.0, 1.0, 0.0, 0.0] and not guess and scale == 0.8      # CFG duplicates the image
    pipe.setup_controlnet([tiny, ControlNetModel(UNetConfig.tiny())])
    assert isinstance(pipe.controlnet, MultiControlNetModel)
    im, keep, guess, scale = pipe.preprocess_controlnet(0.5, [0.0, 0.5], 1.0, [img, img], 64, 64, 2, 1, 1)
    assert len(im) == 2 and keep == [[1.0, 0.0], [1.0, 1.0]] and scale == [0.5, 0.5]
    with pytest.raises(ValueError):
        pipe._controlnet_hook(None, None, None, None, 64, 64, 2, 1, 1, None)


def test_ip_adapter_raw_image_encoding():
    """encode_image / prepare_ip_adapter_image_embeds from raw images (reference model_k_diffusion.py:148-201) with a stand-in
    CLIP vision model: embeddings for ImageProjection layers, penultimate hidden states for the others; CFG stacking"""
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import ImageProjection

    class FakeClipVision(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Linear(3, 8)

        def forward(self, pixel_values, output_hidden_states=False):
            feat = self.p(pixel_values.mean(dim=(2, 3)))                     # [B, 8]
            hs = [feat[:, None, :] * k for k in (1.0, 2.0, 3.0)]              # "hidden states": [-2] is the 2x one
            return types.SimpleNamespace(image_embeds=feat, hidden_states=hs)

    class FakeProcessor:
        def __call__(self, image, return_tensors="pt"):
            return types.SimpleNamespace(pixel_values=torch.from_numpy(np.asarray(image, dtype=np.float32))[None].permute(0, 3, 1, 2))

    unet = UNet2DConditionModel(UNetConfig.tiny()).half()
    enc = FakeClipVision()
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler(), feature_extractor=FakeProcessor(),
                                   image_encoder=enc)
    img = np.random.default_rng(0).random((4, 4, 3)).astype(np.float32)
    with torch.no_grad():
        pos, neg = pipe.encode_image(img, "cpu", 2)
    assert pos.shape == (2, 8) and float(neg.abs().max()) == 0.0 and torch.equal(pos[0], pos[1])
    hp, hn = pipe.encode_image(img, "cpu", 1, output_hidden_states=True)
    assert hp.shape == (1, 1, 8) and torch.allclose(hp[:, 0], 2.0 * pos[:1]) and torch.allclose(hn[:, 0], 2.0 * enc.p.bias[None])
    unet.encoder_hid_proj = types.SimpleNamespace(image_projection_layers=[ImageProjection(8, 64, 4), torch.nn.Identity()])
    out = pipe.prepare_ip_adapter_image_embeds([img, img], None, "cpu", 3, True)
    assert out[0].shape == (6, 1, 8) and float(out[0][:3].detach().abs().max()) == 0.0          # [negative x3 ; positive x3], embeddings
    assert out[1].shape == (6, 1, 1, 8) and torch.allclose(out[1][3:, 0, 0], (2.0 * pos[:1]).expand(3, 8))   # hidden states
    with pytest.raises(ValueError):
        pipe.prepare_ip_adapter_image_embeds([img], None, "cpu", 1, True)
    pipe.image_encoder = None
    with pytest.raises(NotImplementedError):
        pipe.encode_image(img, "cpu", 1)


def test_long_encode_branches_match_reference_goldens():
    """`encode_prompt_function(long_encode=1 / 2)` (reference encoder_prompt_modify.py:395-490 "long prompt weighting",
    :492-689 plain 77-token CLIP) against outputs of the REFERENCE's own functions on the deterministic fake tokenizer /
    encoder (tests/golden/make_golden_prompts.py): embeddings and token ids, with and without clip skip"""
    from inputs import FakeHFClipTokenizer, fake_hf_text_encoder, prompt_encoder_cases
    from diffusionspatialcontrol_amd.modules import encoder_prompt_modify as ep
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "prompt_encoders.npz"))
    pipe = types.SimpleNamespace(tokenizer=FakeHFClipTokenizer(), text_encoder=fake_hf_text_encoder(),
                                 device=torch.device("cpu"), unet=None)
    with torch.no_grad():
        for i, (neg, pos) in enumerate(prompt_encoder_cases()):
            for cs in (None, 2):
                for kind, le in (("long", 1), ("short", 2)):
                    pe, ne, ids = ep.encode_prompt_function(pipe, pos, "cpu", 2, True, neg, clip_skip=cs, long_encode=le)
                    tag = f"{kind}/{i}/skip{cs or 0}"
                    assert np.abs(pe.numpy() - g[tag + "/pos"]).max() < 1e-6 and np.abs(ne.numpy() - g[tag + "/neg"]).max() < 1e-6, tag
                    assert np.array_equal(ids[0], g[tag + "/neg_ids"]) and np.array_equal(ids[1], g[tag + "/pos_ids"]), tag
        pe, ne, ids = ep.encoder_long_prompt(pipe, prompt_encoder_cases()[0][1], "cpu", 1, False)
        assert ne is None and ids[0] is None and np.abs(pe.numpy() - g["long/nocfg/pos"]).max() < 1e-6
    assert ep.parse_prompt_attention("a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).")[1] == ["house", 1.5730000000000004]
    assert ep.parse_prompt_attention("one BREAK two") == [["one BREAK two", 1.0]]          # no BREAK keyword in this variant
    with pytest.raises(ValueError):
        ep.encoder_long_prompt(pipe, ["a", "b"], "cpu", 1, True, ["only one"])


def test_ip_adapter_plus_and_full_projection_conversion():
    """`_convert_ip_adapter_image_proj_to_diffusers` for the Plus (Resampler) and Full checkpoints: a checkpoint written in the
    ORIGINAL key layout from a module's weights converts back to a module computing the same function"""
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import (IPAdapterFullImageProjection,
                                                                            IPAdapterPlusImageProjection)
    torch.manual_seed(0)
    conv = UNet2DConditionLoadersMixin_modify()._convert_ip_adapter_image_proj_to_diffusers
    src = IPAdapterPlusImageProjection(embed_dims=24, output_dims=16, hidden_dims=128, depth=2, dim_head=64, heads=2, num_queries=5,
                                       ffn_ratio=2)
    orig = {"latents": src.latents.data, "proj_in.weight": src.proj_in.weight.data, "proj_in.bias": src.proj_in.bias.data,
            "proj_out.weight": src.proj_out.weight.data, "proj_out.bias": src.proj_out.bias.data,
            "norm_out.weight": src.norm_out.weight.data, "norm_out.bias": src.norm_out.bias.data}
    for i, (ln0, ln1, attn, ff) in enumerate(src.layers):
        orig.update({f"layers.{i}.0.norm1.weight": ln0.weight.data, f"layers.{i}.0.norm1.bias": ln0.bias.data,
                     f"layers.{i}.0.norm2.weight": ln1.weight.data, f"layers.{i}.0.norm2.bias": ln1.bias.data,
                     f"layers.{i}.0.to_q.weight": attn.to_q.weight.data,
                     f"layers.{i}.0.to_kv.weight": torch.cat([attn.to_k.weight.data, attn.to_v.weight.data]),
                     f"layers.{i}.0.to_out.weight": attn.to_out[0].weight.data,
                     f"layers.{i}.1.0.weight": ff[0].weight.data, f"layers.{i}.1.0.bias": ff[0].bias.data,
                     f"layers.{i}.1.1.weight": ff[1].net[0]["proj"].weight.data, f"layers.{i}.1.3.weight": ff[1].net[2].weight.data})
    proj, n_tok = conv(orig)
    x = torch.randn(3, 7, 24)
    with torch.no_grad():
        assert n_tok == 5 and proj(x).shape == (3, 5, 16) and torch.allclose(proj(x), src(x), atol=1e-6)
    full = IPAdapterFullImageProjection(24, 16)
    of = {"proj.0.weight": full.ff.net[0]["proj"].weight.data, "proj.0.bias": full.ff.net[0]["proj"].bias.data,
          "proj.2.weight": full.ff.net[2].weight.data, "proj.2.bias": full.ff.net[2].bias.data,
          "proj.3.weight": full.norm.weight.data, "proj.3.bias": full.norm.bias.data}
    p2, n2 = conv(of)
    with torch.no_grad():
        assert n2 == 257 and torch.allclose(p2(x), full(x), atol=1e-6)
    with pytest.raises(NotImplementedError):
        conv({"norm.weight": torch.zeros(4), "proj.0.weight": torch.zeros(4, 4), "proj.2.weight": torch.zeros(4, 4)})   # FaceID


# ------------------------------------------------------------------ bench.py launcher logic (no GPU needed)
def test_bench_refuses_a_gpus_world_size_mismatch_and_self_launches():
    """`python bench.py --gpus N`: under a launcher a --gpus that differs from WORLD_SIZE is an error that names the right
    command; without one the parent starts N ranks itself (here they fail for lack of a GPU: the parent must relay a
    NON-ZERO status, not 0)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the ranks would really run: the launcher is exercised there by `bench.py --gpus 2` itself")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (r.stderr + r.stdout)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                          # both ranks exit with "bench.py needs an MI355X"
    assert "needs an MI355X" in (r.stderr + r.stdout)


def test_weight_func_key_is_by_value_where_the_callable_allows():
    """a fresh lambda per request (reference app.py:1004) must not force a re-capture: equal code + equal captured values ->
    equal key; different captured values -> different keys; unhashable captures -> object identity"""
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import StableDiffusionPipeline as P

    def make(c):
        return lambda w, s, qk: w * s * c * qk.std()

    assert P._weight_func_key(None) == "default"
    assert P._weight_func_key(lambda w, s, qk: w * s * qk.std()) == "default"
    assert P._weight_func_key(make(2.0)) == P._weight_func_key(make(2.0))
    assert P._weight_func_key(make(2.0)) != P._weight_func_key(make(3.0))
    box = [2.0]
    f1, f2 = (lambda w, s, qk: w * s * box[0] * qk.std()), (lambda w, s, qk: w * s * box[0] * qk.std())
    assert P._weight_func_key(f1) == id(f1) and P._weight_func_key(f2) == id(f2)        # a list cell: by identity


def test_weight_func_key_falls_back_to_identity_for_mutable_captures():
    """round-3 advisor: two closures over ONE mutable object that hashes by identity must not share a captured step (scalars
    read from the object are baked into the capture), and the key must not keep such an object alive"""
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import StableDiffusionPipeline as P

    class Knob:                              # hashable by identity, mutable
        gain = 2.0

    knob = Knob()
    f1, f2 = (lambda w, s, qk: w * s * knob.gain * qk.std()), (lambda w, s, qk: w * s * knob.gain * qk.std())
    assert P._weight_func_key(f1) == id(f1) and P._weight_func_key(f2) == id(f2)
    tup = (1.5, "a", None, (2, True))        # tuples of immutable scalars stay by value

    def make(c):
        return lambda w, s, qk: w * s * c[0] * qk.std()

    assert P._weight_func_key(make(tup)) == P._weight_func_key(make((1.5, "a", None, (2, True))))
    assert isinstance(P._weight_func_key(make(tup)), tuple)


def test_coalesced_requests_table_row_order():
    """txt2img_coalesced: per-request tables [u, c] become ONE table in the batch's row order [u_0..u_{k-1}, c_0..c_{k-1}] (the
    kernels read table row b for batch row b); a request without masks rides along with zeros; mismatched levels are refused"""
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import StableDiffusionPipeline as P
    mk = lambda v: {16: torch.full((2, 16, 77), float(v)), 4: torch.full((2, 4, 77), float(v) + 0.5)}      # noqa: E731
    t0, t1, t2 = mk(1), mk(2), mk(3)
    t1[16][0] += 10                                           # make u and c rows distinguishable
    rs = P._coalesce_region_tables([t0, t1, t2])
    assert sorted(rs) == [4, 16] and rs[16].shape == (6, 16, 77)
    assert [float(rs[16][i, 0, 0]) for i in range(6)] == [1.0, 12.0, 3.0, 1.0, 2.0, 3.0]
    assert [float(rs[4][i, 0, 0]) for i in range(6)] == [1.5, 2.5, 3.5, 1.5, 2.5, 3.5]
    none = torch.FloatTensor(0)                               # encode_region_map's "no table" value
    rs = P._coalesce_region_tables([none, t1])
    assert [float(rs[16][i, 0, 0]) for i in range(4)] == [0.0, 12.0, 0.0, 2.0]
    assert P._coalesce_region_tables([none, none]) is none
    with pytest.raises(ValueError):
        P._coalesce_region_tables([t0, {16: torch.zeros(2, 16, 77)}])
