"""Oracle (test infrastructure only): functional fp32 VAE decoder on a diffusers-named state_dict.

Restates diffusers 0.27.2 `AutoencoderKL.decode` (un-vendored; reached from reference
`source/modules/model_k_diffusion.py:291-299` `decode_latents`) from the published SD1.x VAE structure:
**parity unpinned** (no fixture in the reference can pin it).  `decode_latents` follows :291-299 exactly
(divide by scaling_factor, decode, /2 + 0.5, clamp)."""
import torch
import torch.nn.functional as F


def _c(sd, pre, x, padding=1):
    return F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], padding=padding)


def _gn(sd, pre, x, groups):
    return F.group_norm(x, groups, sd[pre + ".weight"], sd[pre + ".bias"], 1e-6)


def _res(sd, pre, x, groups):
    h = _c(sd, pre + ".conv1", F.silu(_gn(sd, pre + ".norm1", x, groups)))
    h = _c(sd, pre + ".conv2", F.silu(_gn(sd, pre + ".norm2", h, groups)))
    if pre + ".conv_shortcut.weight" in sd:
        x = _c(sd, pre + ".conv_shortcut", x, padding=0)
    return x + h


def vae_decode(sd, z, groups=32):
    sd = {k: v.float() for k, v in sd.items()}
    x = _c(sd, "post_quant_conv", z.float(), padding=0)
    x = _c(sd, "decoder.conv_in", x)
    x = _res(sd, "decoder.mid_block.resnets.0", x, groups)
    b, c, h, w = x.shape
    t = _gn(sd, "decoder.mid_block.attentions.0.group_norm", x, groups).reshape(b, c, h * w).transpose(1, 2)
    p = "decoder.mid_block.attentions.0."
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    o = torch.softmax(q @ k.transpose(1, 2) / (c ** 0.5), dim=-1) @ v
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    x = x + o.transpose(1, 2).reshape(b, c, h, w)
    x = _res(sd, "decoder.mid_block.resnets.1", x, groups)
    i = 0
    while f"decoder.up_blocks.{i}.resnets.0.norm1.weight" in sd:
        j = 0
        while f"decoder.up_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            x = _res(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, groups)
            j += 1
        if f"decoder.up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = _c(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
        i += 1
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, groups))
    return _c(sd, "decoder.conv_out", x)


def decode_latents(sd, latents, scaling_factor=0.18215, groups=32):
    """model_k_diffusion.py:291-299"""
    image = vae_decode(sd, latents.float() / scaling_factor, groups)
    return (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()


def vae_encode_moments(sd, x, groups=32):
    """diffusers 0.27.2 `AutoencoderKL.encode` up to the moments (mean | logvar) tensor (un-vendored, restated from the
    published SD1.x VAE encoder: parity unpinned; reached from reference model_k_diffusion.py:603-606,1234-1246).  The
    down-sampling convolutions pad bottom / right only (`F.pad(x, (0, 1, 0, 1))`, stride 2, no padding)."""
    sd = {k: v.float() for k, v in sd.items()}
    h = _c(sd, "encoder.conv_in", x.float())
    i = 0
    while f"encoder.down_blocks.{i}.resnets.0.norm1.weight" in sd:
        j = 0
        while f"encoder.down_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            h = _res(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, groups)
            j += 1
        pre = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if pre + ".weight" in sd:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[pre + ".weight"], sd[pre + ".bias"], stride=2)
        i += 1
    h = _res(sd, "encoder.mid_block.resnets.0", h, groups)
    b, c, hh, ww = h.shape
    t = _gn(sd, "encoder.mid_block.attentions.0.group_norm", h, groups).reshape(b, c, hh * ww).transpose(1, 2)
    p = "encoder.mid_block.attentions.0."
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    o = torch.softmax(q @ k.transpose(1, 2) / (c ** 0.5), dim=-1) @ v
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    h = h + o.transpose(1, 2).reshape(b, c, hh, ww)
    h = _res(sd, "encoder.mid_block.resnets.1", h, groups)
    h = _c(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, groups)))
    return _c(sd, "quant_conv", h, padding=0)


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution.sample with the noise given: mean + exp(0.5 clamp(logvar, -30, 20)) * noise"""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
