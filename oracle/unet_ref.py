"""Oracle (test infrastructure only): functional fp32 SD-style UNet forward on a diffusers-named state_dict,
and the reference's denoising loop around it.

The UNet blocks are diffusers 0.27.2 code that /root/reference imports but does not vendor
(u_net_condition_modify.py:34-51,352,381,434), restated from their published structure:
**parity unpinned** except for the structural check that the SD1.5 configuration has exactly 859,520,964
parameters.  The loop follows reference `model_k_diffusion.py`: `txt2img` :1027-1050,1043, `model_fn` :1091-1171
(CFG :1162-1166), the sampler call :1175, with `CompVisDenoiser` from oracle.k_diffusion_ref.

`unet_forward` works on ANY state_dict with the diffusers key names, so the product module's weights
(`UNet2DConditionModel.state_dict()`) are the shared input of product and oracle; nothing else is shared.
"""
import math

import torch
import torch.nn.functional as F

from . import k_diffusion_ref as kd
from . import region_attention as ra


def _lin(sd, pre, x):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _conv(sd, pre, x, stride=1, padding=1):
    return F.conv2d(x, sd[pre + ".weight"], sd.get(pre + ".bias"), stride=stride, padding=padding)


def _gn(sd, pre, x, groups, eps):
    return F.group_norm(x, groups, sd[pre + ".weight"], sd[pre + ".bias"], eps)


def _ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _resnet(sd, pre, x, temb, groups, eps):
    h = _conv(sd, pre + ".conv1", F.silu(_gn(sd, pre + ".norm1", x, groups, eps)))
    h = h + _lin(sd, pre + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(sd, pre + ".conv2", F.silu(_gn(sd, pre + ".norm2", h, groups, eps)))
    if pre + ".conv_shortcut.weight" in sd:
        x = _conv(sd, pre + ".conv_shortcut", x, padding=0)
    return x + h


def _attention(sd, pre, x, enc, heads, region_prompt, n_std_groups):
    B, L, C = x.shape
    d = C // heads
    src = x if enc is None else enc
    q = _lin(sd, pre + ".to_q", x).view(B, L, heads, d).transpose(1, 2)
    k = _lin(sd, pre + ".to_k", src).view(B, -1, heads, d).transpose(1, 2)
    v = _lin(sd, pre + ".to_v", src).view(B, -1, heads, d).transpose(1, 2)
    if enc is not None and region_prompt is not None and isinstance(region_prompt["region_state"], dict):
        o = ra.region_attention(q, k, v, region_prompt["region_state"][L], region_prompt["sigma"],
                                n_std_groups=n_std_groups)
    else:
        o = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(d), dim=-1) @ v
    return _lin(sd, pre + ".to_out.0", o.transpose(1, 2).reshape(B, L, C))


def _transformer(sd, pre, x, enc, heads, groups, region_prompt, n_std_groups):
    B, C, h, w = x.shape
    res = x
    x = _gn(sd, pre + ".norm", x, groups, 1e-6)
    linear_proj = sd[pre + ".proj_in.weight"].ndim == 2
    if linear_proj:
        x = _lin(sd, pre + ".proj_in", x.permute(0, 2, 3, 1).reshape(B, h * w, C))
    else:
        x = _conv(sd, pre + ".proj_in", x, padding=0).permute(0, 2, 3, 1).reshape(B, h * w, C)
    i = 0
    while f"{pre}.transformer_blocks.{i}.norm1.weight" in sd:
        bp = f"{pre}.transformer_blocks.{i}"
        x = x + _attention(sd, bp + ".attn1", _ln(sd, bp + ".norm1", x), None, heads, region_prompt, n_std_groups)
        x = x + _attention(sd, bp + ".attn2", _ln(sd, bp + ".norm2", x), enc, heads, region_prompt, n_std_groups)
        y = _lin(sd, bp + ".ff.net.0.proj", _ln(sd, bp + ".norm3", x))
        hid, gate = y.chunk(2, dim=-1)
        x = x + _lin(sd, bp + ".ff.net.2", hid * F.gelu(gate))
        i += 1
    if linear_proj:
        x = _lin(sd, pre + ".proj_out", x).reshape(B, h, w, C).permute(0, 3, 1, 2)
    else:
        x = _conv(sd, pre + ".proj_out", x.reshape(B, h, w, C).permute(0, 3, 1, 2), padding=0)
    return x + res


def t2i_adapter_forward(sd, image, downscale_factor=8):
    """diffusers 0.27.2 `T2IAdapter` (full_adapter; un-vendored, restated from the published structure - parity unpinned):
    pixel-unshuffle, conv_in, four blocks of [2x2 average pool (ceil), 1x1 in_conv when the width changes, ResNets of
    3x3 conv -> ReLU -> 1x1 conv + skip]; returns the four feature maps."""
    sd = {k: v.float() for k, v in sd.items()}
    x = _conv(sd, "adapter.conv_in", F.pixel_unshuffle(image.float(), downscale_factor))
    feats, i = [], 0
    while f"adapter.body.{i}.resnets.0.block1.weight" in sd:
        if i > 0:
            x = F.avg_pool2d(x, 2, 2, ceil_mode=True)
        if f"adapter.body.{i}.in_conv.weight" in sd:
            x = _conv(sd, f"adapter.body.{i}.in_conv", x, padding=0)
        j = 0
        while f"adapter.body.{i}.resnets.{j}.block1.weight" in sd:
            pre = f"adapter.body.{i}.resnets.{j}"
            x = x + _conv(sd, pre + ".block2", F.relu(_conv(sd, pre + ".block1", x)), padding=0)
            j += 1
        feats.append(x)
        i += 1
    return feats


def _encoder_half(sd, cfg, sample, timestep, enc, region_prompt, n_std_groups, add_to_conv_in=None, intrablock=None):
    """time embedding, conv_in (+ an optional additive term: ControlNet's conditioning embedding), down blocks, mid block.
    intrablock: T2I-Adapter features, one per down block (reference u_net_condition_modify.py:1194-1230)"""
    ch, heads_l, G, eps = cfg.block_out_channels, cfg.num_attention_heads, cfg.norm_num_groups, cfg.norm_eps
    half = ch[0] // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = timestep.float().reshape(-1).expand(sample.shape[0])[:, None] * freqs[None]
    temb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)               # flip_sin_to_cos=True, freq_shift=0
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))
    x = _conv(sd, "conv_in", sample.float())
    if add_to_conv_in is not None:
        x = x + add_to_conv_in
    skips = [x]
    intrablock = None if intrablock is None else list(intrablock)
    for i in range(len(ch)):
        extra = intrablock.pop(0) if intrablock else None
        has_attn = f"down_blocks.{i}.attentions.0.norm.weight" in sd
        j = 0
        while f"down_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            x = _resnet(sd, f"down_blocks.{i}.resnets.{j}", x, temb, G, eps)
            if has_attn:
                x = _transformer(sd, f"down_blocks.{i}.attentions.{j}", x, enc, heads_l[i], G, region_prompt, n_std_groups)
                if extra is not None and f"down_blocks.{i}.resnets.{j + 1}.norm1.weight" not in sd:
                    x = x + extra                                   # after the block's last resnet / attention pair
            skips.append(x)
            j += 1
        if f"down_blocks.{i}.downsamplers.0.conv.weight" in sd:
            x = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
        if extra is not None and not has_attn:
            x = x + extra                                           # block without attention: after the whole block
    x = _resnet(sd, "mid_block.resnets.0", x, temb, G, eps)
    if "mid_block.attentions.0.norm.weight" in sd:
        x = _transformer(sd, "mid_block.attentions.0", x, enc, heads_l[-1], G, region_prompt, n_std_groups)
    x = _resnet(sd, "mid_block.resnets.1", x, temb, G, eps)
    return temb, skips, x


def controlnet_forward(sd, cfg, sample, timestep, enc, controlnet_cond, conditioning_scale=1.0, guess_mode=False):
    """diffusers 0.27.2 `ControlNetModel.forward` (un-vendored; the call the reference's model_fn makes,
    model_k_diffusion.py:1134-1142): conditioning embedding (SiLU between its convolutions) added to conv_in's output, the
    UNet encoder half WITHOUT a region prompt (the reference passes no cross_attention_kwargs to the ControlNet), one 1x1
    convolution per skip tensor and for the mid output, times conditioning_scale (guess mode: times logspace(-1, 0)).
    Returns (down residuals, mid residual).  Parity unpinned."""
    sd = {k: v.float() for k, v in sd.items()}
    h = F.silu(_conv(sd, "controlnet_cond_embedding.conv_in", controlnet_cond.float()))
    i = 0
    while f"controlnet_cond_embedding.blocks.{i}.weight" in sd:
        h = F.silu(_conv(sd, f"controlnet_cond_embedding.blocks.{i}", h, stride=2 if i % 2 else 1))
        i += 1
    h = _conv(sd, "controlnet_cond_embedding.conv_out", h)
    _, skips, x = _encoder_half(sd, cfg, sample, timestep, enc.float(), None, 1, add_to_conv_in=h)
    down = [_conv(sd, f"controlnet_down_blocks.{k}", s_, padding=0) for k, s_ in enumerate(skips)]
    mid = _conv(sd, "controlnet_mid_block", x, padding=0)
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        return [d * sc for d, sc in zip(down, scales)], mid * scales[-1]
    return [d * conditioning_scale for d in down], mid * conditioning_scale


def unet_forward(sd, cfg, sample, timestep, enc, region_prompt=None, n_std_groups=1, down_residuals=None, mid_residual=None,
                 intrablock=None):
    """cfg: any object with block_out_channels, num_attention_heads, norm_num_groups, norm_eps.  down_residuals /
    mid_residual: the ControlNet hooks of reference u_net_condition_modify.py:1236-1245,1269-1270."""
    sd = {k: v.float() for k, v in sd.items()}
    ch, heads_l, G, eps = cfg.block_out_channels, cfg.num_attention_heads, cfg.norm_num_groups, cfg.norm_eps
    enc = enc.float()
    temb, skips, x = _encoder_half(sd, cfg, sample, timestep, enc, region_prompt, n_std_groups, intrablock=intrablock)
    if down_residuals is not None:
        skips = [s_ + r for s_, r in zip(skips, down_residuals)]
    if mid_residual is not None:
        x = x + mid_residual
    rev_heads = list(reversed(heads_l))
    for i in range(len(ch)):
        j = 0
        while f"up_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            x = _resnet(sd, f"up_blocks.{i}.resnets.{j}", torch.cat([x, skips.pop()], dim=1), temb, G, eps)
            if f"up_blocks.{i}.attentions.{j}.norm.weight" in sd:
                x = _transformer(sd, f"up_blocks.{i}.attentions.{j}", x, enc, rev_heads[i], G, region_prompt, n_std_groups)
            j += 1
        if f"up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "conv_out", F.silu(_gn(sd, "conv_norm_out", x, G, eps)))


def denoise_loop(sd, cfg, latents, sigmas, text, region_state, guidance_scale, steps_limit=None, on_step=None,
                 sampler=None, sampler_kwargs=None, input_hook=None, v_prediction=False, controlnet=None, adapter=None,
                 extra_input=None):
    """txt2img's loop for n_img images in the row layout [u_0.., c_0..]; returns the final latents (fp32).

    latents: initial noise ALREADY multiplied by sqrt(sigma_0^2 + 1) (model_k_diffusion.py:1043);
    sigmas: the schedule incl. trailing 0 (python floats or a tensor), text: [2*n_img, S, ctx]."""
    # v-prediction: the reference's CompVisVDenoiser.get_v drops cross_attention_kwargs (external_k_diffusion.py:181-182),
    # so its UNet never sees the region prompt
    den = (kd.DiscreteVDenoiser if v_prediction else kd.DiscreteEpsDenoiser)(kd.sd15_alphas_cumprod())
    n_img = latents.shape[0]
    sig = [float(s) for s in sigmas]
    if steps_limit is not None:
        sig = sig[:steps_limit + 1]

    calls = [0]
    seen_sigmas = []
    adapter_seen = []

    def model_fn(x, sigma):
        if input_hook is not None:                     # inpainting: the known region re-imposed on the model input (:1599-1612)
            x = input_hook(x, sigma, calls[0])
        calls[0] += 1
        inp = torch.cat([x] * 2)                                                      # :1097
        if extra_input is not None:                    # 9-channel inpainting UNet: [latents | mask | masked-image latents] (:1617)
            inp = torch.cat([inp, extra_input.float()], dim=1)
        rp = {"region_state": region_state, "sigma": float(sigma[0]), "weight_func": ra.default_weight_func}

        def eps_fn(xin, t, **kw):
            down, mid = None, None
            if controlnet is not None:
                # {"sd", "cond" [2 n_img rows], "scale": one value per DISTINCT sigma in call order}; the ControlNet sees the
                # duplicated latent divided by sqrt(sigma^2 + 1) and the same t (model_k_diffusion.py:1118-1142)
                s_key = float(sigma[0])
                if s_key not in seen_sigmas:
                    seen_sigmas.append(s_key)
                scale = controlnet["scale"][len(seen_sigmas) - 1]
                down, mid = controlnet_forward(controlnet["sd"], cfg, inp / ((sigma[0] ** 2 + 1) ** 0.5), t, text,
                                               controlnet["cond"], scale, controlnet.get("guess_mode", False))
            intra = None
            if adapter is not None:
                # {"state": features duplicated for CFG, "limit": int(len(sigmas) * factor)}: used while fewer than `limit`
                # distinct sigmas have been seen BEFORE this call (model_k_diffusion.py:1109-1117)
                if len(adapter_seen) < adapter["limit"]:
                    intra = adapter["state"]
                if float(sigma[0]) not in adapter_seen:
                    adapter_seen.append(float(sigma[0]))
            return unet_forward(sd, cfg, xin, t, text, region_prompt=None if v_prediction else rp, n_std_groups=n_img,
                                down_residuals=down, mid_residual=mid, intrablock=intra)

        out = den.forward(eps_fn, inp, torch.cat([sigma] * 2))
        return kd.cfg_combine(out, guidance_scale)                                    # :1162-1166

    x = latents.float()
    if sampler is not None:
        # another caller of the same model_fn (k-diffusion call shape `sampler(model, x, sigmas=..., **kw)`): the test
        # hands in the sampler under test, the oracle supplies the fp32 CPU model it drives
        return sampler(model_fn, x, sigmas=torch.tensor(sig, dtype=torch.float32), **(sampler_kwargs or {}))
    old = None
    for i, (a, b, c) in enumerate(kd.dpmpp_2m_coeffs(sig) if sig[-1] == 0 else _coeffs_partial(sig)):
        d = model_fn(x, torch.full((n_img,), sig[i]))
        x = a * x + b * d + (c * old if old is not None else 0.0)
        old = d
        if on_step is not None:
            on_step(i, x)
    return x


def _coeffs_partial(sig):
    """coefficients for a truncated schedule (no trailing zero): same recurrence"""
    full = kd.dpmpp_2m_coeffs(sig + [0.0])
    return full[:len(sig) - 1]
