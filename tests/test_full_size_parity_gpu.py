"""Full-size parity (`-m gpu`): the SD1.5 UNet step and the fused DPM++ 2M loop at the BASELINE.json sizes against the fp32
CPU oracle (oracle/unet_ref.py), through the production kernels (conv3x3 / gemm_tn / LayerNorm fold / flash self-attention /
packed region cross-attention) - the toy-width tests in test_unet_pipeline_gpu.py mostly bypass those.

  * configs[1] (512x512, 2 masks, one image): ONE CFG UNet forward and a 3-step fused loop vs the oracle
    (reference u_net_condition_modify.py:1040-1316, model_k_diffusion.py:1091-1171; ~4.5 s of CPU per oracle step).
  * configs[2] (512x512, 4 masks, 8 images per GPU -> Bc = 16, n_std_groups = 8): image i of the batch equals the
    single-image run (std group = rows {i, 8 + i}), one image against the oracle, and the region
    cross-attention of every level at Bc = 16 against the oracle on the layer's own q / k / v.
  * configs[3] (768x768, batch 8 on one GPU) and configs[4] (SDXL-shape UNet at 1024x1024, 2 images per GPU): the batch
    equals the per-image runs (independent std groups); SDXL's region cross-attention layers against the oracle.
  * generation-to-generation state: tables with more than 32 distinct rows (not compressible) and custom weight_func
    closures must never replay an earlier generation's masks / captured values.
Tolerances are relative to the oracle's output range and written where they are asserted.
"""
import math
import types

import numpy as np
import pytest
import torch

from inputs import FakeTokenizer
from oracle import region_attention as ra
from oracle import unet_ref

pytestmark = pytest.mark.gpu


def _inputs(size, regions, S=77, ctx=768):
    """bench.py's synthetic workload: seed-7 embeddings, phrase r on token columns 2+2r / 3+2r, rectangular
    64-px-aligned masks, UI-default weights"""
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(2, S, ctx, generator=g)
    tok = FakeTokenizer()
    words = [f"object{r}a object{r}b" for r in range(regions)]
    ids = [49406, 320]
    for w in words:
        ids += tok(w).input_ids
    ids = ids + [49407] * (S - len(ids))
    pos = np.array([ids], dtype=np.int64)
    cells = size // 64
    state = {}
    for r, w in enumerate(words):
        m = np.full((size, size), 255, dtype=np.uint8)
        x0, x1 = (r * cells) // regions, ((r + 1) * cells) // regions
        m[(cells // 4) * 64:(3 * cells // 4) * 64, x0 * 64:x1 * 64] = 0
        state[w] = {"map": m, "weight": 0.5, "mask_outsides": 0.0}
    return emb, [pos.copy(), pos], state, tok


@pytest.fixture(scope="module")
def sd15():
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
    cfg = UNetConfig.sd15()
    torch.manual_seed(0)
    with torch.device("cuda"):
        unet = UNet2DConditionModel(cfg)
    unet = unet.half().eval()
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}      # fp16-representable, shared with the oracle
    pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    return types.SimpleNamespace(cfg=cfg, unet=unet, sd=sd, pipe=pipe)


def _latent(i, n=64):
    return torch.randn(4, n, n, generator=torch.Generator().manual_seed(1000 + i))


def _region_tables(pipe, state, size, ids):
    from diffusionspatialcontrol_amd.modules.encode_region_map_function import encode_region_map
    return encode_region_map(pipe, state, size, size, 1, text_ids=ids)


def test_sd15_unet_forward_full_size_vs_oracle(sd15):
    """configs[1], one CFG forward: fp16 HIP path vs fp32 oracle on shared weights.  Tolerance 4e-3 of the output range
    (max), 5e-4 (mean): ~60 fp16 layers deep; the toy-width test allows 1e-2."""
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    x = torch.stack([_latent(0), _latent(0)]).half()
    t = torch.tensor([731.25, 731.25])
    text = emb.half()
    rp = {"region_state": rs, "sigma": torch.tensor([4.0], device="cuda"), "weight_func": lambda w, s, qk: w * s * qk.std()}
    with torch.no_grad():
        out = sd15.unet(x.cuda(), t.cuda(), text.cuda(), cross_attention_kwargs={"region_prompt": rp}).sample.float().cpu()
    ref = unet_ref.unet_forward(sd15.sd, sd15.cfg, x.float(), t, text.float(),
                                region_prompt={"region_state": rs, "sigma": 4.0, "weight_func": None})
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    print(f"full-size forward: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.3f}")
    assert err.max().item() < 4e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() < 5e-4 * scale, (err.mean().item(), scale)


def _oracle_loop(sd15, lat, sig, text_rows, rs, steps):
    return unet_ref.denoise_loop(sd15.sd, sd15.cfg, lat.float() * math.sqrt(sig[0] ** 2 + 1), sig, text_rows.float(), rs, 7.5,
                                 steps_limit=steps)


def _fused(sd15, lat, sig, text_rows, rs, steps):
    import inspect
    pipe = sd15.pipe
    wf = inspect.signature(pipe.txt2img).parameters["weight_func"].default
    sg = torch.tensor(sig[:steps + 1]).half().cuda()
    x0 = lat.half().cuda() * (sg[0] ** 2 + 1) ** 0.5                      # txt2img's start (model_k_diffusion.py:1043)
    return pipe._denoise_fused(x0, sg, text_rows.half().cuda(), rs, wf, 7.5, lat.shape[0], {}, -1, 0).float().cpu()


def test_sd15_three_step_loop_full_size_vs_oracle(sd15):
    """configs[1]: the first 3 of the 25 DPM++ 2M Karras steps, fused loop (captured UNet step + dsc_cfg_dpmpp2m_step) vs
    the oracle loop on the same fp16-rounded schedule.  Tolerance 2e-3 of the latent range (bench.py observed 1.1e-3)."""
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    lat = _latent(0)[None]
    text = torch.cat([emb[0:1], emb[1:2]])
    ref = _oracle_loop(sd15, lat.half(), sig, text.half(), rs, 3)
    got = _fused(sd15, lat, sig, text, rs, 3)
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    print(f"3-step loop: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.2f}")
    assert err.max().item() < 2e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() < 3e-4 * scale


def test_config1_single_step_one_mask(sd15):
    """BASELINE configs[0] as written: ONE denoise step of a 64x64 latent with ONE region mask.  The 1-mask table comes from
    `encode_region_map` (reference encode_region_map_function.py:21-77: one phrase -> token columns 2, 3; every level's table
    has two distinct rows), the step is one CFG `model_fn` call (model_k_diffusion.py:1091-1171, CompVisDenoiser.forward
    external_k_diffusion.py:109-114) + the first DPM++ 2M update to sigma_1; fused HIP loop vs the fp32 oracle loop.
    Bounds: the per-forward tolerance of this file carried through one update - 2e-3 of the latent range (max), 3e-4 (mean);
    observed 5e-4 / 8e-5.
    Also checked: the denoised estimate of that single model call (a sampler-free view of the same step: steps_limit = 1 on a
    [sigma_0, 0] schedule returns the CFG-combined denoised itself).  Its bound is wider BY CONSTRUCTION: denoised = x - sigma_0 *
    (-6.5 eps_u + 7.5 eps_c) multiplies the two forwards' errors by sigma_0 = 14.6 and by ~10 (the CFG combination), so a forward at
    its usual 1.1e-3 of the eps range shows as 3.2e-3 of the latent range here (observed) - bound 6e-3 max, 1e-3 mean; the first
    DPM++ 2M update then weights it with 1 - sigma_1 / sigma_0 = 0.16."""
    emb, ids, state, _ = _inputs(512, 1)
    assert len(state) == 1
    rs = _region_tables(sd15.pipe, state, 512, ids)
    assert sorted(rs) == [64, 256, 1024, 4096]
    for L, w in rs.items():                                  # one mask: background row + region row (columns 2, 3 carry 0.5)
        assert w.shape == (2, L, 77) and len(torch.unique(w[1], dim=0)) == 2
        assert torch.count_nonzero(w[..., 4:]).item() == 0 and w[1, :, 2:4].max().item() == 0.5
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    lat = _latent(0)[None]
    text = torch.cat([emb[0:1], emb[1:2]])
    for name, schedule, tol_max, tol_mean in (("one DPM++ 2M step", sig, 2e-3, 3e-4), ("denoised of one model call", [sig[0], 0.0], 6e-3, 1e-3)):
        ref = _oracle_loop(sd15, lat.half(), schedule, text.half(), rs, 1)
        got = _fused(sd15, lat, schedule, text, rs, 1)
        scale = ref.abs().max().item()
        err = (got - ref).abs()
        print(f"configs[0] {name}: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.2f}")
        assert torch.isfinite(got).all()
        assert err.max().item() < tol_max * scale, (name, err.max().item(), scale)
        assert err.mean().item() < tol_mean * scale, (name, err.mean().item(), scale)


def test_config3_eight_images_four_masks(sd15):
    """BASELINE configs[2] on one GPU: 8 images per generation, 4 region masks (Bc = 16, n_std_groups = 8)."""
    from diffusionspatialcontrol_amd.modules.attention_modify import AttnProcessor2_0
    emb, ids, state, _ = _inputs(512, 4)
    pipe = sd15.pipe
    rs1 = _region_tables(pipe, state, 512, ids)
    n = 8
    lats = torch.stack([_latent(i) for i in range(n)]).half()
    kw = dict(height=512, width=512, num_inference_steps=25, guidance_scale=7.5, output_type="latent", region_map_state=state,
              sampler_name="sample_dpmpp_2m", sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2].half().cuda(),
              negative_prompt_embeds=emb[0:1].half().cuda(), text_input_ids=ids)
    # (1) the product entry point at the full batch: finite, and the 3-step truncation below is the same code path
    sig = pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    text8 = torch.cat([emb[0:1].repeat(n, 1, 1), emb[1:2].repeat(n, 1, 1)])
    rs8 = {L: w.repeat(n, 1, 1) for L, w in rs1.items()}                    # encode_region_map's .repeat(num_images, 1, 1) (:122)
    got8 = _fused(sd15, lats, sig, text8, rs8, 3)
    assert torch.isfinite(got8).all()
    full = pipe.txt2img(None, latents=lats.cuda(), num_images_per_prompt=n, **kw)[0]      # all 25 steps through txt2img itself
    assert full.shape == (n, 4, 64, 64) and torch.isfinite(full).all()
    # round 4: the 25 steps at the batch of 8 are CHECKED, not only finite - image 0 of the batch against the oracle's own 25-step
    # run of that image with the 4-mask tables (~2 min of CPU): the north star's stated tolerance (8e-3 max / 1e-3 mean of range)
    text1_ = torch.cat([emb[0:1], emb[1:2]])
    ref25 = _oracle_loop(sd15, lats[0:1], sig, text1_.half(), rs1, 25)
    e25 = (full[0:1].float().cpu() - ref25).abs()
    sc25 = ref25.abs().max().item()
    print(f"configs[2], image 0 of 8 after 25 steps vs oracle: max/range {e25.max().item() / sc25:.2e} mean/range {e25.mean().item() / sc25:.2e}")
    assert e25.max().item() < FINAL_LATENT_TOL_MAX * sc25 and e25.mean().item() < FINAL_LATENT_TOL_MEAN * sc25
    # (2) image i of the batch == the single-image run (its std group is rows {i, 8 + i}); different launch geometry
    # (grids, split-K counts) -> equal to rounding: 2e-3 of the latent range
    text1 = torch.cat([emb[0:1], emb[1:2]])
    scale = got8.abs().max().item()
    for i in (0, 5):
        single = _fused(sd15, lats[i:i + 1], sig, text1, rs1, 3)
        d = (single[0] - got8[i]).abs().max().item()
        print(f"image {i}: batch-of-8 vs single {d:.3e} (range {scale:.2f})")
        assert d < 2e-3 * scale, (i, d, scale)
    # (3) one image of the 4-mask workload against the oracle loop
    ref = _oracle_loop(sd15, lats[5:6], sig, text1.half(), rs1, 3)
    e = (got8[5] - ref[0]).abs()
    print(f"image 5 of 8 vs oracle: max {e.max().item():.3e} mean {e.mean().item():.3e}")
    assert e.max().item() < 2.5e-3 * ref.abs().max().item()
    # (4) region cross-attention of each level at Bc = 16 against the oracle, on the layer's own q / k / v
    seen = {}

    class Recorder(AttnProcessor2_0):
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, region_prompt=None):
            out = super().__call__(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, region_prompt=region_prompt)
            L = hidden_states.shape[1]
            if encoder_hidden_states is not None and L not in seen:
                q = attn.to_q(hidden_states)
                k, v = attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)
                B, _, C = q.shape
                H = attn.heads
                seen[L] = (q.view(B, L, H, C // H), k.view(B, 77, H, C // H), v.view(B, 77, H, C // H), attn.to_out[0], out)
            return out

    old = dict(sd15.unet.attn_processors)
    sd15.unet.set_attn_processor(Recorder())
    try:
        x = torch.cat([lats, lats]).cuda()
        t = torch.full((2 * n,), 640.5, device="cuda")
        rp = {"region_state": rs8, "sigma": torch.tensor([6.0], device="cuda"), "weight_func": lambda w, s, qk: w * s * qk.std(),
              "n_std_groups": n}
        with torch.no_grad():
            y = sd15.unet(x, t, text8.half().cuda(), cross_attention_kwargs={"region_prompt": rp}).sample
    finally:
        sd15.unet.set_attn_processor(old)
    assert torch.isfinite(y).all() and sorted(seen) == [64, 256, 1024, 4096]
    for L, (q, k, v, to_out, got) in seen.items():
        for i in (0, 3, 7):
            rows = [i, n + i]                                                  # image i's (uncond, cond) rows
            qc, kc, vc = (z[rows].float().cpu().transpose(1, 2) for z in (q, k, v))
            exp = ra.region_attention(qc, kc, vc, rs8[L][rows], 6.0).transpose(1, 2).reshape(2, L, -1)
            exp = torch.nn.functional.linear(exp, to_out.weight.float().cpu(), to_out.bias.float().cpu())
            err = (got[rows].float().cpu() - exp).abs()
            assert err.max().item() < 8e-3 * max(1.0, exp.abs().max().item()), (L, i, err.max().item())


# ------------------------------------------------------------------ generation-to-generation state of the captured step
def _tiny_pipe(seed=0):
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
    torch.manual_seed(seed)
    cfg = UNetConfig.tiny()
    unet = UNet2DConditionModel(cfg).half().cuda()
    return cfg, StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())


def _random_tables(seed, distinct):
    """{L: [2, L, 77]} with `distinct` different rows per level (distinct > 32: not compressible)"""
    g = torch.Generator().manual_seed(seed)
    rs = {}
    for L in (256, 64, 16, 4):
        base = torch.zeros(distinct, 77)
        base[:, 2:8] = torch.rand(distinct, 6, generator=g)
        idx = torch.randint(0, distinct, (L,), generator=g)
        idx[:min(L, distinct)] = torch.arange(min(L, distinct))
        rs[L] = base[idx][None].repeat(2, 1, 1).contiguous()
    return rs


def _run_tiny(pipe, cfg, rs, wf, seed=11):
    g = torch.Generator().manual_seed(seed)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).half().cuda()
    lat = torch.randn(1, 4, 16, 16, generator=g).half().cuda()
    sig = pipe.get_sigmas(4, {"scheduler": "karras"}).half().cuda()
    return pipe._denoise_fused(lat * (sig[0] ** 2 + 1) ** 0.5, sig, text, rs, wf, 7.5, 1, {}, -1, 0).float().cpu()


def test_incompressible_tables_are_refreshed_between_generations():
    """Two generations with the same shapes but different tables of > 32 distinct rows: the second must equal a FRESH
    pipeline's result (the captured step reads static dense buffers refreshed in place, not the first generation's tensors)."""
    wf = lambda w, s, qk: w * s * qk.std()                                     # noqa: E731
    rs_a, rs_b = _random_tables(1, 48), _random_tables(2, 48)
    from diffusionspatialcontrol_amd import ops
    assert ops.compress_region_table(rs_a[256]) is None and ops.compress_region_table(rs_b[64]) is None
    cfg, pipe = _tiny_pipe()
    first = _run_tiny(pipe, cfg, rs_a, wf)
    second = _run_tiny(pipe, cfg, rs_b, wf)
    _, fresh_pipe = _tiny_pipe()
    fresh = _run_tiny(fresh_pipe, cfg, rs_b, wf)
    scale = fresh.abs().max().item()
    assert (first - second).abs().max().item() > 1e-3 * scale                  # the tables matter
    assert (second - fresh).abs().max().item() < 2e-2 * scale, (second - fresh).abs().max().item()   # toy widths: MIOpen atomics
    # and back to a compressible table on the same pipeline
    rs_c = _random_tables(3, 5)
    third = _run_tiny(pipe, cfg, rs_c, wf)
    fresh_c = _run_tiny(_tiny_pipe()[1], cfg, rs_c, wf)
    assert (third - fresh_c).abs().max().item() < 2e-2 * fresh_c.abs().max().item()


def test_custom_weight_func_closures_are_not_shared_between_generations():
    """Two closures of ONE code object with different captured values (a custom, non-default weight_func): each generation
    must run its own callable - the captured step is keyed by the callable object, not by its code."""
    def make(gain):
        return lambda w, s, qk: w * s * qk.std() * gain

    rs = _random_tables(4, 6)
    cfg, pipe = _tiny_pipe()
    one = _run_tiny(pipe, cfg, rs, make(1.0))                                 # gain 1: the default function in disguise
    three = _run_tiny(pipe, cfg, rs, make(3.0))
    fresh_three = _run_tiny(_tiny_pipe()[1], cfg, rs, make(3.0))
    scale = fresh_three.abs().max().item()
    assert (one - three).abs().max().item() > 1e-3 * scale
    assert (three - fresh_three).abs().max().item() < 2e-2 * scale


# ------------------------------------------------------------------ BASELINE configs[3] / [4] at their per-GPU batch
def test_config4_768_eight_images(sd15):
    """BASELINE configs[3]: SD1.5 at 768x768 (L = 9216 / 2304 / 576 / 144), 2 region masks, batch 8 on one GPU (Bc = 16,
    n_std_groups = 8): two fused steps, finite, and image i of the batch equals its single-image run to 2e-3 of the latent
    range (the oracle comparison of this geometry is test_config4_sd15_768_step's, layer by layer)."""
    emb, ids, state, _ = _inputs(768, 2)
    rs1 = _region_tables(sd15.pipe, state, 768, ids)
    assert sorted(rs1) == [144, 576, 2304, 9216]
    n = 8
    lats = torch.stack([_latent(i, 96) for i in range(n)]).half()
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    text8 = torch.cat([emb[0:1].repeat(n, 1, 1), emb[1:2].repeat(n, 1, 1)])
    rs8 = {L: w.repeat(n, 1, 1) for L, w in rs1.items()}
    got8 = _fused(sd15, lats, sig, text8, rs8, 2)
    assert got8.shape == (n, 4, 96, 96) and torch.isfinite(got8).all()
    scale = got8.abs().max().item()
    text1 = torch.cat([emb[0:1], emb[1:2]])
    for i in (0, 6):
        single = _fused(sd15, lats[i:i + 1], sig, text1, rs1, 2)
        d = (single[0] - got8[i]).abs().max().item()
        print(f"768x768 image {i}: batch-of-8 vs single {d:.3e} (range {scale:.2f})")
        assert d < 2e-3 * scale, (i, d, scale)
    # round 4: ... and one image OF THE BATCH against the oracle loop at this geometry (two steps, ~20 s of CPU): the batch-of-8
    # launch geometry (Bc = 16: other grids, split counts, 128-column GEMM tiles) inside the per-step tolerance of the 512x512 loop
    ref = _oracle_loop(sd15, lats[6:7], sig, text1.half(), rs1, 2)
    e = (got8[6] - ref[0]).abs()
    print(f"768x768 image 6 of 8 vs oracle (2 steps): max {e.max().item():.3e} mean {e.mean().item():.3e} range {ref.abs().max().item():.2f}")
    assert e.max().item() < 2e-3 * ref.abs().max().item() and e.mean().item() < 3e-4 * ref.abs().max().item()


def test_config5_sdxl_two_images_per_gpu():
    """BASELINE configs[4]: SDXL-base UNet geometry at 1024x1024, 16 images over 8 GPUs = 2 per GPU (Bc = 4, n_std_groups = 2).
    One CFG forward: each image's rows equal that image's own Bc = 2 forward (independent std groups), and the region
    cross-attention of both attention levels (L = 4096, 10 heads; L = 1024, 20 heads; d = 64, context 2048) is checked
    against the oracle on the layer's own q / k / v with the per-image groups.  (The reference has no SDXL pipeline,
    SURVEY.md 8d: this extrapolates the same processor contract.)"""
    from diffusionspatialcontrol_amd.modules.attention_modify import AttnProcessor2_0
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
    torch.manual_seed(0)
    with torch.device("cuda"):
        unet = UNet2DConditionModel(UNetConfig.sdxl_base()).half().eval()
    g = torch.Generator().manual_seed(6)
    n = 2
    x1 = torch.randn(n, 4, 128, 128, generator=g).half()
    enc1 = torch.randn(2, 77, 2048, generator=g).half()
    x = torch.cat([x1, x1]).cuda()                                          # rows [u_0, u_1, c_0, c_1]
    enc = torch.cat([enc1[0:1].repeat(n, 1, 1), enc1[1:2].repeat(n, 1, 1)]).cuda()
    t = torch.full((2 * n,), 400.0, device="cuda")
    rs = {}
    gg = torch.Generator().manual_seed(2)
    for L in (16384, 4096, 1024):
        w = torch.zeros(2, L, 77)
        w[:, torch.rand(L, generator=gg) < 0.3, 2:4] = 0.5
        w[:, torch.rand(L, generator=gg) < 0.3, 4:6] += 0.5
        rs[L] = w
    rs_n = {L: w.repeat(n, 1, 1) for L, w in rs.items()}
    wf = lambda w_, s_, qk: w_ * s_ * qk.std()                                # noqa: E731
    seen = {}

    class Recorder(AttnProcessor2_0):
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, region_prompt=None):
            out = super().__call__(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, region_prompt=region_prompt)
            L = hidden_states.shape[1]
            if encoder_hidden_states is not None and L not in seen and hidden_states.shape[0] == 2 * n:
                q = attn.to_q(hidden_states)
                k, v = attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)
                B, _, C = q.shape
                H = attn.heads
                seen[L] = (q.view(B, L, H, C // H), k.view(B, 77, H, C // H), v.view(B, 77, H, C // H), attn.to_out[0], out)
            return out

    unet.set_attn_processor(Recorder())
    with torch.no_grad():
        rp = {"region_state": rs_n, "sigma": torch.tensor([4.0], device="cuda"), "weight_func": wf, "n_std_groups": n}
        y = unet(x, t, enc, cross_attention_kwargs={"region_prompt": rp}).sample
        assert y.shape == (2 * n, 4, 128, 128) and torch.isfinite(y).all()
        scale = y.float().abs().max().item()
        for i in range(n):
            rp1 = {"region_state": rs, "sigma": torch.tensor([4.0], device="cuda"), "weight_func": wf}
            y1 = unet(x[[i, n + i]], t[:2], enc[[i, n + i]], cross_attention_kwargs={"region_prompt": rp1}).sample
            dmax = (y1.float() - y[[i, n + i]].float()).abs().max().item()
            print(f"SDXL image {i}: rows of the Bc=4 forward vs its own Bc=2 forward {dmax:.3e} (range {scale:.2f})")
            assert dmax < 4e-3 * scale, (i, dmax, scale)
    assert sorted(seen) == [1024, 4096]
    for L, (q, k, v, to_out, got) in seen.items():
        for i in range(n):
            rows = [i, n + i]
            qc, kc, vc = (z[rows].float().cpu().transpose(1, 2) for z in (q, k, v))
            exp = ra.region_attention(qc, kc, vc, rs_n[L][rows], 4.0).transpose(1, 2).reshape(2, L, -1)
            exp = torch.nn.functional.linear(exp, to_out.weight.float().cpu(), to_out.bias.float().cpu())
            err = (got[rows].float().cpu() - exp).abs()
            assert err.max().item() < 8e-3 * max(1.0, exp.abs().max().item()), (L, i, err.max().item())


def test_step_is_recaptured_when_the_tuning_profile_changes():
    """a slot's captured step keeps the launch rules it was captured under; after ops.set_tuning_profile the next generation
    re-captures (st["profile"]) and gives the same latents (the profiles' kernels give equal bytes,
    test_conv3x3_and_gemm_profiles_give_equal_bytes; the toy widths run MIOpen's atomic convolutions, hence a tolerance here)"""
    from diffusionspatialcontrol_amd import ops
    wf = lambda w, s, qk: w * s * qk.std()                                     # noqa: E731
    rs = _random_tables(3, 8)
    cfg, pipe = _tiny_pipe()
    try:
        ops.set_tuning_profile("latency")
        a = _run_tiny(pipe, cfg, rs, wf)
        st_a = next(iter(pipe._graphs.values()))
        assert st_a["profile"] == "latency"
        b = _run_tiny(pipe, cfg, rs, wf)
        assert next(iter(pipe._graphs.values())) is st_a                        # same shapes, same profile: the same capture
        ops.set_tuning_profile("throughput")
        c = _run_tiny(pipe, cfg, rs, wf)
        st_c = next(iter(pipe._graphs.values()))
        assert st_c is not st_a and st_c["profile"] == "throughput" and len(pipe._graphs) == 1
    finally:
        ops.set_tuning_profile("latency")
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() < 2e-2 * scale and (a - c).abs().max().item() < 2e-2 * scale


def test_time_embedding_rows_from_the_schedule_table(sd15):
    """The fused loop computes every step's time-embedding projections once per schedule (UNet.temb_add_table) and the sampler
    kernels hand the coming step's row to the captured step (ops.USE_TEMB_HOIST): a table row equals what the forward computes
    for that timestep, and the loop's latents equal the ones of the per-step form bit for bit."""
    import inspect
    from diffusionspatialcontrol_amd import ops
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    sigmas = torch.tensor(sig[:5]).half()
    wf = inspect.signature(sd15.pipe.txt2img).parameters["weight_func"].default
    text = torch.cat([emb[0:1], emb[1:2]]).half()
    x0 = _latent(0)[None].half().cuda() * (sigmas[0].cuda() ** 2 + 1) ** 0.5
    unet = sd15.unet
    ts = torch.tensor([999.0, 500.25, 3.0, 41.5, 0.0, 123.0, 77.7, 8.0, 640.0], device="cuda")
    tab = unet.temb_add_table(ts)
    assert tab.shape == (9, unet.temb_width())
    for i in (0, 4, 8):                                                        # rows of a 9-row table vs a 2-row forward
        probe = torch.empty((2, 1), device="cuda", dtype=torch.float16)
        want = unet._temb_all(unet._time_act(probe, ts[i:i + 1]))
        assert torch.equal(tab[i], want[0]) and torch.equal(tab[i], want[1])
    outs = {}
    saved = ops.USE_TEMB_HOIST
    try:
        for hoist in (True, False):
            ops.USE_TEMB_HOIST = hoist
            pipe = StableDiffusionPipeline(None, None, FakeTokenizer(), unet, SD15Scheduler())
            outs[hoist] = pipe._denoise_fused(x0, sigmas.cuda(), text.cuda(), rs, wf, 7.5, 1, {}, -1, 0).float().cpu()
            st = next(iter(pipe._graphs.values()))
            assert (st["tadd"] is not None) == hoist
    finally:
        ops.USE_TEMB_HOIST = saved
    assert torch.isfinite(outs[True]).all() and torch.equal(outs[True], outs[False])


def test_shared_cfg_prefix_equals_the_full_batch(sd15):
    """forward(cfg_shared_prefix=True): with [x; x] rows and one timestep the layers in front of the first cross-attention run
    once per image; the result equals the full-batch forward up to the launch geometry of those layers (other grids, other
    split counts), and the fused loop with it stays inside the oracle tolerance (test_sd15_three_step_loop_full_size_vs_oracle
    runs with the default, which is on)."""
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    from diffusionspatialcontrol_amd.modules.attention_modify import AttnProcessor2_0
    unet = sd15.unet
    for n in (1, 3):
        lat = torch.stack([_latent(i) for i in range(n)]).half().cuda()
        x = torch.cat([lat, lat]) * 0.07
        t = torch.full((2 * n,), 700.0, device="cuda")
        text = torch.cat([emb[0:1].repeat(n, 1, 1), emb[1:2].repeat(n, 1, 1)]).half().cuda()
        rp = {"region_state": {L: w.repeat(n, 1, 1) for L, w in rs.items()}, "sigma": torch.tensor([5.0], device="cuda"),
              "weight_func": lambda w, s, qk: w * s * qk.std(), "n_std_groups": n}
        with torch.no_grad():
            full = unet(x, t, text, cross_attention_kwargs={"region_prompt": rp}).sample
            shared = unet(x, t, text, cross_attention_kwargs={"region_prompt": rp}, cfg_shared_prefix=True).sample
        assert shared.shape == full.shape and torch.isfinite(shared).all()
        scale = full.float().abs().max().item()
        err = (shared.float() - full.float()).abs()
        # two fp16 evaluations of the same function through different launch geometries: ~half an fp16 ulp of the output on average
        assert err.max().item() < 4e-3 * scale and err.mean().item() < 6e-4 * scale, (n, err.max().item(), err.mean().item(), scale)
        assert (shared.float() - full.float()).abs().max().item() > 0 or n == 0      # (not the same launches: a real second path)


# ------------------------------------------------------------------ end-to-end: all 25 steps, the north star's stated tolerance
# Observed on MI355X (round 3; printed by the test): see DESIGN.md section 2 for the numbers these bounds were derived from.
FINAL_LATENT_TOL_MAX = 8e-3            # max |final latent - oracle| / oracle range, after 25 DPM++ 2M Karras steps (observed 2.4e-3)
FINAL_LATENT_TOL_MEAN = 1e-3           # mean |...| / range (observed 2.4e-4)


@pytest.mark.parametrize("profile", ["latency", "throughput"])
def test_sd15_25_step_loop_full_size_vs_oracle(sd15, profile):
    """configs[1] END TO END: all 25 DPM++ 2M Karras steps at 512x512, 2 region masks, seed-1000 latent, through `txt2img`'s
    fused loop (captured UNet step + dsc_cfg_dpmpp2m_step) under BOTH launch-rule profiles, final latents vs the fp32 CPU
    oracle loop (oracle/unet_ref.denoise_loop; the value reference `sampler(model_fn, latents, ...)` returns,
    model_k_diffusion.py:1175).  The bound is the north star's "stated fp16 tolerance" (DESIGN.md section 2)."""
    from diffusionspatialcontrol_amd import ops
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    lat = _latent(0)[None]
    text = torch.cat([emb[0:1], emb[1:2]])
    ref = _oracle_final_latents(sd15, lat, sig, text, rs)
    try:
        ops.set_tuning_profile(profile)
        got = _fused(sd15, lat, sig, text, rs, 25)
    finally:
        ops.set_tuning_profile("latency")
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    print(f"25-step loop [{profile}]: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.3f} "
          f"(max/range {err.max().item() / scale:.2e}, mean/range {err.mean().item() / scale:.2e})")
    assert err.max().item() < FINAL_LATENT_TOL_MAX * scale, (err.max().item(), scale)
    assert err.mean().item() < FINAL_LATENT_TOL_MEAN * scale, (err.mean().item(), scale)


def _requests(n):
    """n DIFFERENT batch-1 requests at 512x512: request 0 = configs[1]'s (2 masks, seed-7 embeddings, seed-1000 latent); the others
    their own embeddings, mask count (1, 4, 2, ...) and latent"""
    reqs = []
    for i in range(n):
        emb, ids, state, _ = _inputs(512, (2, 1, 4, 2, 1, 4, 2, 1)[i % 8])
        if i:
            emb = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(70 + i))
        reqs.append({"prompt_embeds": emb[1:2].half(), "negative_prompt_embeds": emb[0:1].half(), "text_input_ids": ids,
                     "region_map_state": state, "latents": _latent(i)[None].half()})
    return reqs


def test_coalesced_requests_equal_their_single_runs(sd15):
    """Serving mode `txt2img_coalesced`: THREE different concurrent requests (2 / 1 / 4 masks, own prompts and latents) in one
    captured step per sigma.  Image i of the coalesced run equals request i's own `txt2img` call - its std group is rows {i, 3 + i},
    its tables ride on those rows - up to launch-geometry rounding (other grids / split counts at Bc = 6): 2e-3 of the latent range
    after 3 steps, the bound of the batch-vs-single checks above; and request 0, whose inputs are configs[1]'s, is within the
    3-step oracle bound."""
    pipe = sd15.pipe
    reqs = _requests(3)
    kw = dict(height=512, width=512, num_inference_steps=25, guidance_scale=7.5, output_type="latent", sampler_opt={"scheduler": "karras"})
    sig = pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    # 3-step truncation through the same entry point: a 25-step schedule cut after 3 steps (the schedule is an argument of the loop)
    cut = {"orig": pipe._schedule}

    def first3(steps, params, device, dtype):
        s = cut["orig"](steps, params, device, dtype)[:4].clone()
        return s

    pipe._schedule = first3
    try:
        got = [o.float().cpu() for o in pipe.txt2img_coalesced([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in r.items()} for r in reqs], **kw)]
        singles = []
        for r in reqs:
            singles.append(pipe.txt2img(None, latents=r["latents"].cuda(), region_map_state=r["region_map_state"],
                                        sampler_name="sample_dpmpp_2m", prompt_embeds=r["prompt_embeds"].cuda(),
                                        negative_prompt_embeds=r["negative_prompt_embeds"].cuda(), text_input_ids=r["text_input_ids"],
                                        **kw)[0].float().cpu())
    finally:
        del pipe._schedule                                      # back to the class's method
    scale = max(s_.abs().max().item() for s_ in singles)
    for i, (g_, s_) in enumerate(zip(got, singles)):
        d = (g_ - s_).abs().max().item()
        print(f"coalesced request {i}: vs its own txt2img call {d:.3e} (range {scale:.2f})")
        assert g_.shape == (1, 4, 64, 64) and torch.isfinite(g_).all()
        assert d < 2e-3 * scale, (i, d, scale)
    assert (got[0] - got[1]).abs().max().item() > 0.1 * scale          # (the requests ARE different)
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(pipe, state, 512, ids)
    ref = _oracle_loop(sd15, _latent(0)[None].half(), sig, torch.cat([emb[0:1], emb[1:2]]).half(), rs, 3)
    e = (got[0] - ref).abs()
    print(f"coalesced request 0 vs oracle (3 steps): max {e.max().item():.3e} mean {e.mean().item():.3e}")
    assert e.max().item() < 2.5e-3 * ref.abs().max().item()


def test_coalesced_requests_beyond_32_distinct_table_rows():
    """Nine coalesced requests whose region tables together hold MORE than 32 distinct rows per level (the prepared-operand kernels'
    limit): the step must take the dense-table route (static buffers refreshed in place) and every request must still equal its own
    `txt2img` call.  Toy-width UNet at 128x128 (levels 256 / 64 / 16 / 4), four one-cell masks per request with request-specific
    weights -> 4 distinct rows per request and level, 36 in the batch."""
    cfg, pipe = _tiny_pipe(3)
    tok = FakeTokenizer()
    g = torch.Generator().manual_seed(21)
    reqs = []
    for i in range(9):
        words = [f"object{r}a object{r}b" for r in range(4)]
        ids = [49406, 320]
        for w in words:
            ids += tok(w).input_ids
        ids = ids + [49407] * (77 - len(ids))
        pos = np.array([ids], dtype=np.int64)
        state = {}
        for r, w in enumerate(words):
            m = np.full((128, 128), 255, dtype=np.uint8)
            m[(r // 2) * 64:(r // 2 + 1) * 64, (r % 2) * 64:(r % 2 + 1) * 64] = 0
            state[w] = {"map": m, "weight": 0.3 + 0.05 * i + 0.01 * r, "mask_outsides": 0.0}
        emb = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).half()
        reqs.append({"prompt_embeds": emb[1:2].cuda(), "negative_prompt_embeds": emb[0:1].cuda(), "text_input_ids": [pos.copy(), pos],
                     "region_map_state": state, "latents": torch.randn(1, 4, 16, 16, generator=g).half().cuda()})
    kw = dict(height=128, width=128, num_inference_steps=4, guidance_scale=7.5, output_type="latent", sampler_opt={"scheduler": "karras"})
    from diffusionspatialcontrol_amd.modules.encode_region_map_function import encode_region_map
    per = [encode_region_map(pipe, r["region_map_state"], 128, 128, 1, text_ids=r["text_input_ids"]) for r in reqs]
    merged = pipe._coalesce_region_tables(per)
    assert len(torch.unique(merged[256].reshape(-1, 77), dim=0)) > 32 and pipe._compress_tables(merged) is None      # the dense route
    got = pipe.txt2img_coalesced(reqs, **kw)
    scale = max(o.float().abs().max().item() for o in got)
    for i, r in enumerate(reqs):
        single = pipe.txt2img(None, latents=r["latents"], region_map_state=r["region_map_state"], sampler_name="sample_dpmpp_2m",
                              prompt_embeds=r["prompt_embeds"], negative_prompt_embeds=r["negative_prompt_embeds"],
                              text_input_ids=r["text_input_ids"], **kw)[0]
        d = (got[i].float() - single.float()).abs().max().item()
        assert torch.isfinite(got[i]).all() and d < 4e-3 * scale, (i, d, scale)
    assert (got[0].float() - got[1].float()).abs().max().item() > 1e-2 * scale


def test_coalesced_pair_25_steps_image0_vs_oracle(sd15):
    """configs[1]'s request coalesced with a second, different request (k = 2): all 25 DPM++ 2M Karras steps; image 0 against the
    fp32 oracle's 25-step latents of the ONE-image run - the stated end-to-end tolerance (8e-3 max / 1e-3 mean of range) holds for
    a request that shared its steps with another one."""
    reqs = _requests(2)
    emb, ids, state, _ = _inputs(512, 2)
    rs = _region_tables(sd15.pipe, state, 512, ids)
    sig = sd15.pipe.get_sigmas(25, {"scheduler": "karras"}).half().float().tolist()
    ref = _oracle_final_latents(sd15, _latent(0)[None], sig, torch.cat([emb[0:1], emb[1:2]]), rs)
    got = sd15.pipe.txt2img_coalesced([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in r.items()} for r in reqs], height=512, width=512,
                                      num_inference_steps=25, guidance_scale=7.5, output_type="latent",
                                      sampler_opt={"scheduler": "karras"})
    g0 = got[0].float().cpu()
    scale = ref.abs().max().item()
    err = (g0 - ref).abs()
    print(f"coalesced pair, image 0 after 25 steps: max/range {err.max().item() / scale:.2e} mean/range {err.mean().item() / scale:.2e}")
    assert torch.isfinite(got[1]).all()
    assert err.max().item() < FINAL_LATENT_TOL_MAX * scale, (err.max().item(), scale)
    assert err.mean().item() < FINAL_LATENT_TOL_MEAN * scale, (err.mean().item(), scale)


_ORACLE_25 = {}


def _oracle_final_latents(sd15, lat, sig, text, rs):
    """the oracle's 25-step latents, computed once per session (~2 min of CPU) and shared by the two profile cases"""
    if "ref" not in _ORACLE_25:
        _ORACLE_25["ref"] = _oracle_loop(sd15, lat.half(), sig, text.half(), rs, 25)
    return _ORACLE_25["ref"]


def test_sd15_768_forward_full_size_vs_oracle(sd15):
    """configs[3] geometry, one CFG UNet forward at 768x768 (L = 9216 / 2304 / 576 / 144; the 12x12 level runs the ragged
    convolution tiles) vs the fp32 oracle on shared weights.  Same bound as the 512x512 forward: 4e-3 of range, 5e-4 mean."""
    emb, ids, state, _ = _inputs(768, 2)
    rs = _region_tables(sd15.pipe, state, 768, ids)
    x = torch.stack([_latent(0, 96), _latent(0, 96)]).half()
    t = torch.tensor([540.5, 540.5])
    text = emb.half()
    rp = {"region_state": rs, "sigma": torch.tensor([2.5], device="cuda"), "weight_func": lambda w, s, qk: w * s * qk.std()}
    with torch.no_grad():
        out = sd15.unet(x.cuda(), t.cuda(), text.cuda(), cross_attention_kwargs={"region_prompt": rp}).sample.float().cpu()
    ref = unet_ref.unet_forward(sd15.sd, sd15.cfg, x.float(), t, text.float(),
                                region_prompt={"region_state": rs, "sigma": 2.5, "weight_func": None})
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    print(f"768x768 forward: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.3f}")
    assert err.max().item() < 4e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() < 5e-4 * scale, (err.mean().item(), scale)


def test_vae_decode_full_size_vs_oracle():
    """SURVEY 8f rank 1 at the SD1.x geometry: 64x64 latent -> 512x512 RGB through the production kernels (conv3x3, GroupNorm,
    1x1 GEMMs, and the single 512-dim attention head over 4096 tokens on the HIP path: gemm_tn scores -> dsc_softmax_rows ->
    gemm_tn P.V) vs the fp32 oracle (oracle/vae_ref.py; reference model_k_diffusion.py:291-299) on shared weights.
    Bound 1.5e-2 of the output range (max), 2e-3 (mean) - the toy-width test's bound."""
    from diffusionspatialcontrol_amd.modules.vae_decoder import AutoencoderKLDecoder
    from oracle import vae_ref
    torch.manual_seed(5)
    vae = AutoencoderKLDecoder().half().eval()
    sd = {k: v.clone() for k, v in vae.state_dict().items()}
    vae = vae.cuda()
    z = (torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(11)) * 0.9).half()
    seen = {}
    import diffusionspatialcontrol_amd.modules.vae_decoder as vd
    orig = vd.VaeAttention._attend
    def spy(self, q, k, vt):
        seen["hip"] = True
        return orig(self, q, k, vt)
    vd.VaeAttention._attend = spy
    try:
        with torch.no_grad():
            out = vae.decode(z.cuda()).sample.float().cpu()
    finally:
        vd.VaeAttention._attend = orig
    assert seen.get("hip"), "the VAE attention did not take the HIP path"
    with torch.no_grad():
        ref = vae_ref.vae_decode(sd, z.float())
    assert out.shape == (1, 3, 512, 512)
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    print(f"VAE decode 64x64 -> 512x512: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.3f}")
    assert err.max().item() < 1.5e-2 * scale and err.mean().item() < 2e-3 * scale, (err.max().item(), err.mean().item(), scale)


def test_sdxl_shape_forward_full_size_vs_oracle():
    """configs[4] geometry, ONE CFG forward of the SDXL-base-shaped UNet at 1024x1024 (3 levels, transformer depth 0 / 2 / 10,
    5 / 10 / 20 heads of dim 64, context 2048, linear projections: 140 attention layers) against the fp32 oracle on shared
    weights - the whole forward, not only its attention layers.  (The reference has no SDXL pipeline, SURVEY.md 8d: the same
    processor contract on an SDXL-shaped UNet.)  Bound: 4e-3 of the output range (max), 5e-4 (mean), as for SD1.5 (observed 1.1e-3 / 1.5e-4)."""
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
    cfg = UNetConfig.sdxl_base()
    torch.manual_seed(0)
    with torch.device("cuda"):
        unet = UNet2DConditionModel(cfg)
    unet = unet.half().eval()
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    x1 = torch.randn(1, 4, 128, 128, generator=g).half()
    x = torch.cat([x1, x1])
    enc = torch.randn(2, 77, 2048, generator=g).half()
    t = torch.tensor([400.0, 400.0])
    rs = {}
    gg = torch.Generator().manual_seed(2)
    for L in (16384, 4096, 1024):
        w = torch.zeros(2, L, 77)
        w[:, torch.rand(L, generator=gg) < 0.3, 2:4] = 0.5
        w[:, torch.rand(L, generator=gg) < 0.3, 4:6] += 0.5
        rs[L] = w
    rp = {"region_state": rs, "sigma": torch.tensor([4.0], device="cuda"), "weight_func": lambda w_, s_, qk: w_ * s_ * qk.std()}
    with torch.no_grad():
        out = unet(x.cuda(), t.cuda(), enc.cuda(), cross_attention_kwargs={"region_prompt": rp}).sample.float().cpu()
    del unet
    torch.cuda.empty_cache()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, cfg, x.float(), t, enc.float(), region_prompt={"region_state": rs, "sigma": 4.0, "weight_func": None})
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    print(f"SDXL-shape 1024x1024 forward: max {err.max().item():.3e} mean {err.mean().item():.3e} range {scale:.3f}")
    assert err.max().item() < 4e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() < 5e-4 * scale, (err.mean().item(), scale)
