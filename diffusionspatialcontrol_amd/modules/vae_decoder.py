"""VAE decoder (latents -> RGB) - the step right after the denoising loop, SURVEY.md 8f rank 1 - and encoder
(RGB -> latent moments) - the step right before it in img2img / inpainting (reference model_k_diffusion.py:600-606,1234-1246).

Counterpart of what reference `source/modules/model_k_diffusion.py` reaches through `self.vae.decode(latents)` in
`decode_latents` (:291-299) / `latent_to_image` (:533-539): diffusers 0.27.2 `AutoencoderKL` (un-vendored), decoder half
only.  Structure restated from the published SD1.x VAE (diffusers key names, so `load_state_dict` of the `decoder.*` and
`post_quant_conv.*` entries of an AutoencoderKL checkpoint works): post_quant_conv 1x1 -> conv_in 4->512 -> mid
(ResNet, single-head attention over h*w tokens, ResNet) -> 4 up blocks of 3 ResNets (512, 512, 256, 128 channels, nearest
x2 upsample + conv after the first three) -> GroupNorm + SiLU -> conv_out 128->3.  **Parity unpinned**: diffusers is
absent; the oracle restatement (oracle/vae_ref.py) shares only the weights.

Device work: channels-last activations; GroupNorm(+SiLU) in libdsc_hip.so (`dsc_groupnorm_silu_nhwc`), 1x1 convs as
token-major GEMMs, 3x3 convolutions on dsc_conv3x3_nhwc_f16; the single 512-dim attention head (d > 160: too wide for
the flash kernel's registers) runs GEMM (scores) -> dsc_softmax_rows_f16 -> GEMM (p.v) on the hand-written kernels.
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .u_net_condition_modify import Conv1x1, GroupNormAct, _image, _tokens


@dataclass
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2                 # the decoder uses layers_per_block + 1 ResNets per up block
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def tiny():
        return VaeConfig(block_out_channels=(32, 32, 64, 64), norm_num_groups=8)


def _conv(conv, x, residual=None, upsample=False):
    """3x3 / pad 1 convolution of a decoder block: dsc_conv3x3_nhwc_f16 (bias / residual / nearest-2x upsample fused) where
    it covers the shape, else the library convolution"""
    if ops.conv3x3_supported(x, conv.weight, upsample=upsample):
        return ops.conv3x3(x, conv.weight, conv.bias, residual=residual, upsample=upsample)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    h = conv(x)
    return h if residual is None else ops.add_bias_residual(residual, h)


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = GroupNormAct(groups, cin, 1e-6, act=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = GroupNormAct(groups, cout, 1e-6, act=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv1x1(cin, cout) if cin != cout else None

    def forward(self, x):
        h = _conv(self.conv1, self.norm1(x))
        if self.conv_shortcut is None:
            return _conv(self.conv2, self.norm2(h), residual=x)          # skip add in the convolution's epilogue
        return self.conv_shortcut(x, residual=_conv(self.conv2, self.norm2(h)))


class VaeAttention(nn.Module):
    """diffusers `Attention(c, heads=1, dim_head=c, norm_num_groups=g, residual_connection=True, bias=True)`"""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = GroupNormAct(groups, c, 1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def _hip_covers(self, t):
        """the GEMM -> row softmax -> GEMM form takes fp16 GPU tokens whose counts fit the GEMM kernel's tiles"""
        b, L, c = t.shape
        return t.is_cuda and t.dtype == torch.float16 and L % 64 == 0 and c % 64 == 0 and L <= 16384

    def _attend(self, q, k, vt):
        """one image: q, k [L, c], vt = v^T [c, L] -> softmax(q k^T / sqrt(c)) v  [L, c].  The scores are materialised once
        in fp16 (dsc_linear_f16: 'weight' = the keys), normalised in fp32 (dsc_softmax_rows_f16) and contracted with v^T as
        the second GEMM's [N, K] operand - a 512-channel head does not fit the flash kernel's registers."""
        # q arrives pre-multiplied by c^-1/2 (forward: the scale is folded into the to_q projection), so the fp16 scores are the
        # SCALED logits - the headroom the replaced SDPA path had (unscaled q.k^T at c = 512 is 22.6x larger and could reach inf)
        scores = ops.linear(q, k, prefer_kernel=True)                                    # [L, L] = (q c^-1/2) . k^T
        probs = ops.softmax_rows(scores, scale=1.0, out=scores)                          # in place: one 2 L^2-byte buffer
        return ops.linear(probs, vt, prefer_kernel=True)                                 # [L, c] = p . v

    def forward(self, x):
        b, c, h, w = x.shape
        t = _tokens(self.group_norm(x))
        if self._hip_covers(t):
            from .u_net_condition_modify import _derived
            wq, bq = _derived(self, "to_q_scaled", (self.to_q.weight, self.to_q.bias),
                              lambda: ((self.to_q.weight.detach().float() * c ** -0.5).to(self.to_q.weight.dtype).contiguous(),
                                       (self.to_q.bias.detach().float() * c ** -0.5).to(self.to_q.bias.dtype).contiguous()))
            q = ops.linear(t, wq, bq, prefer_kernel=True)                                # = to_q(t) / sqrt(c)
            k = ops.linear(t, self.to_k.weight, self.to_k.bias, prefer_kernel=True)
            L = h * w
            # v^T [c, L] straight from a GEMM with the roles swapped (x = W_v, 'weight' = the tokens); its bias is a ROW
            # constant there, so it rides as the GEMM's residual operand (a [c, L] expansion of b_v, cached per L)
            # (ONE cached expansion: a new image size replaces the previous one, the cache does not grow with the sizes seen)
            bcol = _derived(self, "vbias", (self.to_v.bias, L),
                            lambda: self.to_v.bias.detach()[:, None].expand(c, L).contiguous())
            outs = []
            for i in range(b):
                vt = ops.linear(self.to_v.weight, t[i], residual=bcol, prefer_kernel=True)
                outs.append(self._attend(q[i], k[i], vt))
            o = outs[0][None] if b == 1 else torch.stack(outs)
        else:
            q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
            o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]    # one head of dim c (library call)
        return _image(ops.linear(o, self.to_out[0].weight, self.to_out[0].bias, residual=_tokens(x)), h, w)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(c, c, groups), VaeResnet(c, c, groups)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])


class _Up(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if upsample:
            self.upsamplers = nn.ModuleList([nn.ModuleDict({"conv": nn.Conv2d(cout, cout, 3, padding=1)})])


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch = list(reversed(cfg.block_out_channels))
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = _Mid(ch[0], cfg.norm_num_groups)
        self.up_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.up_blocks.append(_Up(prev, c, cfg.layers_per_block + 1, cfg.norm_num_groups, i < len(ch) - 1))
            prev = c
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, ch[-1], 1e-6, act=True)
        self.conv_out = nn.Conv2d(ch[-1], cfg.out_channels, 3, padding=1)


class _Cfg(dict):
    __getattr__ = dict.get


class AutoencoderKLDecoder(nn.Module):
    """`vae.decode(latents).sample` of the reference pipeline (decoder half of AutoencoderKL)."""

    def __init__(self, cfg: VaeConfig = None):
        super().__init__()
        cfg = cfg or VaeConfig()
        self.cfg = cfg
        self.config = _Cfg(scaling_factor=cfg.scaling_factor, block_out_channels=cfg.block_out_channels)
        self.post_quant_conv = Conv1x1(cfg.latent_channels, cfg.latent_channels)
        self.decoder = Decoder(cfg)
        self._channels_last = False

    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    def decode(self, z, return_dict=True):
        d = self.decoder
        if not self._channels_last:
            self.to(memory_format=torch.channels_last)
            for conv in (d.conv_in, d.conv_out):                 # 4 / 3 channels: no NHWC igemm in MIOpen, run NCHW
                conv.weight.data = conv.weight.data.contiguous()
            self._channels_last = True
        b, c, h, w = z.shape
        z = F.conv2d(z.contiguous(), self.post_quant_conv.weight, self.post_quant_conv.bias)
        x = d.conv_in(z).contiguous(memory_format=torch.channels_last)
        x = d.mid_block.resnets[0](x)
        x = d.mid_block.attentions[0](x)
        x = d.mid_block.resnets[1](x)
        for blk in d.up_blocks:
            for res in blk.resnets:
                x = res(x)
            if hasattr(blk, "upsamplers"):
                x = _conv(blk.upsamplers[0]["conv"], x, upsample=True)
        hn = d.conv_norm_out(x)
        wcl = d.conv_out.weight.contiguous(memory_format=torch.channels_last)
        if ops.conv3x3_supported(hn, wcl):
            x = ops.conv3x3(hn, wcl, d.conv_out.bias, out_nchw=True)          # 128 -> 3 channels, channel-major image
        else:
            x = d.conv_out(hn.contiguous()).contiguous()
        return type("DecoderOutput", (), {"sample": x})() if return_dict else (x,)


# --------------------------------------------------------------------------------------------------- encoder half
class _Down(nn.Module):
    def __init__(self, cin, cout, n, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if downsample:     # diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then a stride-2 / pad-0 convolution
            self.downsamplers = nn.ModuleList([nn.ModuleDict({"conv": nn.Conv2d(cout, cout, 3, stride=2, padding=0)})])


class Encoder(nn.Module):
    """diffusers 0.27.2 `Encoder` of AutoencoderKL (un-vendored; structure of the published SD1.x VAE): conv_in 3 -> 128,
    four down blocks of 2 ResNets (128, 256, 512, 512; a stride-2 convolution after the first three), mid (ResNet,
    single-head attention, ResNet), GroupNorm + SiLU, conv_out 512 -> 2 * latent_channels (mean | logvar)."""

    def __init__(self, cfg):
        super().__init__()
        ch = list(cfg.block_out_channels)
        self.conv_in = nn.Conv2d(cfg.out_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.down_blocks.append(_Down(prev, c, cfg.layers_per_block, cfg.norm_num_groups, i < len(ch) - 1))
            prev = c
        self.mid_block = _Mid(ch[-1], cfg.norm_num_groups)
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, ch[-1], 1e-6, act=True)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """diffusers `DiagonalGaussianDistribution` (what `vae.encode(x).latent_dist` is): mean | logvar moments, logvar
    clamped to [-30, 20]; `sample(generator)` draws on the generator's device like diffusers' randn_tensor."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(AutoencoderKLDecoder):
    """Both halves, diffusers key names (`encoder.*`, `quant_conv.*`, `decoder.*`, `post_quant_conv.*`): 83,653,863
    parameters in the SD1.x configuration (the published AutoencoderKL total).  `encode(x).latent_dist` as the reference's
    img2img (:603-606) and `_encode_vae_image` (:1234-1246) use it.  **Parity unpinned** (diffusers absent); oracle
    restatement on shared weights: oracle/vae_ref.py `vae_encode`."""

    def __init__(self, cfg: VaeConfig = None):
        super().__init__(cfg)
        cfg = self.cfg
        self.config["latent_channels"] = cfg.latent_channels
        self.encoder = Encoder(cfg)
        self.quant_conv = Conv1x1(2 * cfg.latent_channels, 2 * cfg.latent_channels)

    @property
    def device(self):
        return self.quant_conv.weight.device

    def encode(self, x, return_dict=True):
        e = self.encoder
        cl = torch.channels_last
        x = x.to(self.dtype)
        wt = _derived_w(e.conv_in, "fewcin", lambda w: w.reshape(w.shape[0], -1).t().contiguous())
        if x.is_cuda and x.dtype == torch.float16 and x.shape[1] <= 16:
            h = ops.conv3x3_fewcin(x, wt, e.conv_in.bias, e.conv_in.out_channels)          # NCHW image in, channels-last out
        else:
            h = e.conv_in(x).contiguous(memory_format=cl)
        for blk in e.down_blocks:
            for res in blk.resnets:
                h = res(h)
            if hasattr(blk, "downsamplers"):
                conv = blk.downsamplers[0]["conv"]
                wcl = _derived_w(conv, "cl", lambda w: w.contiguous(memory_format=cl))
                if h.shape[-1] % 2 == 0 and h.shape[-2] % 2 == 0 and ops.conv3x3_supported(h, wcl):
                    h = ops.conv3x3(h, wcl, conv.bias, stride2_pad_br=True)                # odd pixels of the stride-1 taps
                else:
                    h = F.conv2d(F.pad(h, (0, 1, 0, 1)), conv.weight, conv.bias, stride=2).contiguous(memory_format=cl)
        h = e.mid_block.resnets[0](h)
        h = e.mid_block.attentions[0](h)
        h = e.mid_block.resnets[1](h)
        hn = e.conv_norm_out(h)
        wcl = _derived_w(e.conv_out, "cl", lambda w: w.contiguous(memory_format=cl))
        if ops.conv3x3_supported(hn, wcl):
            m = ops.conv3x3(hn, wcl, e.conv_out.bias, out_nchw=True)                        # 512 -> 8 channels, channel-major
        else:
            m = e.conv_out(hn).contiguous()
        m = F.conv2d(m, self.quant_conv.weight, self.quant_conv.bias)                       # 8 x 8 1x1 on the moments
        dist = DiagonalGaussianDistribution(m)
        return type("AutoencoderKLOutput", (), {"latent_dist": dist})() if return_dict else (dist,)


def _derived_w(conv, key, build):
    """weight of `conv` in another layout, cached on the module and rebuilt when the parameter changes"""
    from .u_net_condition_modify import _derived
    return _derived(conv, "w_" + key, (conv.weight,), lambda: build(conv.weight))
