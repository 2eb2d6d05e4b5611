"""SD-style conditional UNet for the MI355X hot path - the build's own counterpart of
reference `source/modules/u_net_condition_modify.py` (`UNet2DConditionModel`, forward :1040-1316), whose
blocks live in the un-vendored diffusers 0.27.2 (SURVEY.md Appendix B gives the structure this file mirrors).

What is kept from the reference surface:
  * class name `UNet2DConditionModel`, `forward(sample, timestep, encoder_hidden_states, ...,
    cross_attention_kwargs=..., down_block_additional_residuals=..., mid_block_additional_residual=...).sample`
  * `attn_processors` / `set_attn_processor` (:689-749) over `Attention` submodules that call
    `processor(attn, hidden_states, encoder_hidden_states=..., attention_mask=..., **cross_attention_kwargs)`
  * diffusers parameter names (`down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight`, ...), so a
    diffusers SD1.5 / SDXL-base UNet state_dict loads with `load_state_dict`
  * `UNet2DConditionLoadersMixin_modify`, the symbol the reference imports but never defines (:23)

Device work: activations are kept channels-last (NHWC) end to end - the same bytes as the transformer's token-major
[B, h*w, C] view - so MIOpen's NHWC implicit-GEMM convolutions and the hipBLASLt projection GEMMs (plain library
GEMMs, reached through torch) need no layout transposes; GroupNorm+SiLU(+time-embedding add), GEGLU,
self-attention and region cross-attention run in libdsc_hip.so (see ..ops).
"""
import inspect
import math
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .attention_modify import AttnProcessor, AttnProcessor2_0


class ImageProjection(nn.Module):
    """diffusers 0.27.2 `ImageProjection` [recalled; package absent - parity unpinned]: CLIP image embedding
    [B, image_embed_dim] -> `num_image_text_embeds` context tokens [B, T, cross_attention_dim] (Linear + LayerNorm)."""

    def __init__(self, image_embed_dim=768, cross_attention_dim=768, num_image_text_embeds=32):
        super().__init__()
        self.num_image_text_embeds = num_image_text_embeds
        self.image_embeds = nn.Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        b = image_embeds.shape[0]
        return self.norm(self.image_embeds(image_embeds).reshape(b, self.num_image_text_embeds, -1))


class _GeluFeedForward(nn.Module):
    """diffusers `FeedForward(dim, dim_out, mult, activation_fn="gelu", bias=...)`: net.0.proj (Linear + exact GELU), net.1
    Dropout(0), net.2 Linear - the key names of the IP-Adapter Full / Plus image projections"""

    def __init__(self, dim, dim_out, mult, bias=True):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.ModuleList([nn.ModuleDict({"proj": nn.Linear(dim, inner, bias=bias)}), nn.Dropout(0.0),
                                  nn.Linear(inner, dim_out, bias=bias)])

    def forward(self, x):
        return self.net[2](F.gelu(self.net[0]["proj"](x)))


class IPAdapterFullImageProjection(nn.Module):
    """diffusers 0.27.2 `IPAdapterFullImageProjection` [recalled; package absent - parity unpinned]: CLIP penultimate hidden
    states [B, T, image_embed_dim] -> LayerNorm(FeedForward(x)) [B, T, cross_attention_dim]"""

    def __init__(self, image_embed_dim=1024, cross_attention_dim=1024):
        super().__init__()
        self.ff = _GeluFeedForward(image_embed_dim, cross_attention_dim, mult=1)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        return self.norm(self.ff(image_embeds))


class _ResamplerAttention(nn.Module):
    """diffusers `Attention(query_dim, dim_head, heads, out_bias=False)` as the Plus resampler uses it: no biases, queries
    from the latents, keys / values from [image tokens ; latents]"""

    def __init__(self, dim, dim_head, heads):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=False), nn.Dropout(0.0)])

    def forward(self, latents, context):
        b, n, _ = latents.shape
        sp = lambda t: t.reshape(b, t.shape[1], self.heads, -1).transpose(1, 2)  # noqa: E731
        o = F.scaled_dot_product_attention(sp(self.to_q(latents)), sp(self.to_k(context)), sp(self.to_v(context)))
        return self.to_out[0](o.transpose(1, 2).reshape(b, n, -1))


class IPAdapterPlusImageProjection(nn.Module):
    """diffusers 0.27.2 `IPAdapterPlusImageProjection` (the Perceiver-style Resampler of IP-Adapter Plus) [recalled; parity
    unpinned]: `num_queries` learned latents attend to the projected image tokens and themselves for `depth` layers."""

    def __init__(self, embed_dims=768, output_dims=1024, hidden_dims=1280, depth=4, dim_head=64, heads=16, num_queries=8,
                 ffn_ratio=4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, num_queries, hidden_dims) / hidden_dims ** 0.5)
        self.proj_in = nn.Linear(embed_dims, hidden_dims)
        self.proj_out = nn.Linear(hidden_dims, output_dims)
        self.norm_out = nn.LayerNorm(output_dims)
        self.layers = nn.ModuleList([nn.ModuleList([
            nn.LayerNorm(hidden_dims), nn.LayerNorm(hidden_dims), _ResamplerAttention(hidden_dims, dim_head, heads),
            nn.Sequential(nn.LayerNorm(hidden_dims), _GeluFeedForward(hidden_dims, hidden_dims, ffn_ratio, bias=False))])
            for _ in range(depth)])

    def forward(self, x):
        latents = self.latents.repeat(x.size(0), 1, 1)
        x = self.proj_in(x)
        for ln0, ln1, attn, ff in self.layers:
            residual = latents
            latents = ln1(latents)
            latents = attn(latents, torch.cat([ln0(x), latents], dim=-2)) + residual
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))


class MultiIPAdapterImageProjection(nn.Module):
    """diffusers 0.27.2 `MultiIPAdapterImageProjection` [recalled]: one projection layer per loaded IP-Adapter; takes the
    list of per-adapter image embeddings ([B, num_images, D], or the deprecated single [B, D] tensor) and returns the
    list of per-adapter token tensors the IP-Adapter processors consume."""

    def __init__(self, layers):
        super().__init__()
        self.image_projection_layers = nn.ModuleList(layers)

    def forward(self, image_embeds):
        if not isinstance(image_embeds, (list, tuple)):
            image_embeds = [image_embeds.unsqueeze(1)]
        if len(image_embeds) != len(self.image_projection_layers):
            raise ValueError(f"image_embeds must have the same length as image_projection_layers, got {len(image_embeds)} "
                             f"and {len(self.image_projection_layers)}")
        out = []
        for e, layer in zip(image_embeds, self.image_projection_layers):
            b, n = e.shape[0], e.shape[1]
            t = layer(e.reshape((b * n,) + tuple(e.shape[2:])))
            out.append(t.reshape((b, n * t.shape[1]) + tuple(t.shape[2:])))      # [B, num_images * T, ctx]
        return out


class UNet2DConditionLoadersMixin_modify:
    """The symbol reference `u_net_condition_modify.py:23,70` imports but never defines; the reference's IP-Adapter
    loader calls `_load_ip_adapter_weights` on it (ip_adapter.py:231) and expects the extra (FaceID) LoRAs back."""

    def _convert_ip_adapter_image_proj_to_diffusers(self, state_dict):
        """(projection module, number of image tokens) from an IP-Adapter checkpoint's `image_proj` dict, by its keys [diffusers
        0.27.2 `_convert_ip_adapter_image_proj_to_diffusers`, recalled]: "proj.weight" -> standard (Linear + LayerNorm, 4 tokens);
        "proj.3.weight" -> Full (MLP + LayerNorm on the CLIP hidden states, 257 tokens); "latents" -> Plus (Resampler).  FaceID
        ("norm.weight" + "proj.0/2" with LoRA weights) is not built."""
        if "proj.3.weight" in state_dict:
            proj = IPAdapterFullImageProjection(image_embed_dim=state_dict["proj.0.weight"].shape[0],
                                                cross_attention_dim=state_dict["proj.3.weight"].shape[0])
            ren = {"proj.0": "ff.net.0.proj", "proj.2": "ff.net.2", "proj.3": "norm"}
            proj.load_state_dict({next((k.replace(a_, b_) for a_, b_ in ren.items() if k.startswith(a_)), k): v
                                  for k, v in state_dict.items()})
            return proj, 257
        if "latents" in state_dict:
            hidden = state_dict["latents"].shape[2]
            proj = IPAdapterPlusImageProjection(embed_dims=state_dict["proj_in.weight"].shape[1],
                                                output_dims=state_dict["proj_out.weight"].shape[0], hidden_dims=hidden,
                                                heads=state_dict["layers.0.0.to_q.weight"].shape[0] // 64,
                                                num_queries=state_dict["latents"].shape[1],
                                                ffn_ratio=state_dict["layers.0.1.1.weight"].shape[0] / hidden,
                                                depth=1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("layers.")))
            sd = {}
            for k, v in state_dict.items():
                # original Resampler layout: layers.{i}.0 = PerceiverAttention (norm1, norm2, to_q, to_kv, to_out),
                # layers.{i}.1 = FeedForward Sequential (0 LayerNorm, 1 Linear, 2 GELU, 3 Linear)
                n = k.replace("0.to", "2.to").replace("1.0.weight", "3.0.weight").replace("1.0.bias", "3.0.bias")
                n = n.replace("1.1.weight", "3.1.net.0.proj.weight").replace("1.3.weight", "3.1.net.2.weight")
                if "norm1" in n:
                    sd[n.replace("0.norm1", "0")] = v
                elif "norm2" in n:
                    sd[n.replace("0.norm2", "1")] = v
                elif "to_kv" in n:
                    kk, vv = v.chunk(2, dim=0)
                    sd[n.replace("to_kv", "to_k")], sd[n.replace("to_kv", "to_v")] = kk, vv
                elif "to_out" in n:
                    sd[n.replace("to_out", "to_out.0")] = v
                else:
                    sd[n] = v
            proj.load_state_dict(sd)
            return proj, state_dict["latents"].shape[1]
        if "proj.weight" not in state_dict:
            raise NotImplementedError("IP-Adapter FaceID image projections (and their LoRA weights) are not built")
        num_tokens = 4
        w = state_dict["proj.weight"]
        ctx = w.shape[0] // num_tokens
        proj = ImageProjection(image_embed_dim=w.shape[1], cross_attention_dim=ctx, num_image_text_embeds=num_tokens)
        proj.load_state_dict({"image_embeds.weight": w, "image_embeds.bias": state_dict["proj.bias"],
                              "norm.weight": state_dict["norm.weight"], "norm.bias": state_dict["norm.bias"]})
        return proj, num_tokens

    def _load_ip_adapter_weights(self, state_dicts, low_cpu_mem_usage=False):
        """Installs IPAdapterAttnProcessor2_0 on every cross-attention layer (self-attention keeps AttnProcessor2_0) and
        the image projection(s) as `encoder_hid_proj` [diffusers 0.27.2 `UNet2DConditionLoadersMixin` behaviour, recalled].
        state_dicts: one {"image_proj": {...}, "ip_adapter": {"<id>.to_k_ip.weight", "<id>.to_v_ip.weight"}} per adapter;
        ids 1, 3, 5, ... number the cross-attention layers in the order down_blocks, up_blocks, mid_block (the
        registration order of diffusers' UNet, which the published IP-Adapter checkpoints follow)."""
        from .attention_modify import AttnProcessor2_0, IPAdapterAttnProcessor2_0
        if not isinstance(state_dicts, list):
            state_dicts = [state_dicts]
        projs, num_tokens = [], []
        for sd in state_dicts:
            pr, nt = self._convert_ip_adapter_image_proj_to_diffusers(sd["image_proj"])
            projs.append(pr)
            num_tokens.append(nt)
        order = [(n, m) for pre in ("down_blocks", "up_blocks", "mid_block")
                 for n, m in self.named_modules() if isinstance(m, Attention) and n.startswith(pre)]
        key_id = 1
        ref = next(self.parameters())
        for name, attn in order:
            if not attn.is_cross_attention:
                attn.set_processor(AttnProcessor2_0())
                continue
            proc = IPAdapterAttnProcessor2_0(hidden_size=attn.inner_dim, cross_attention_dim=self.cfg.cross_attention_dim,
                                             num_tokens=num_tokens, scale=1.0)
            sd = {}
            for i, s in enumerate(state_dicts):
                sd[f"to_k_ip.{i}.weight"] = s["ip_adapter"][f"{key_id}.to_k_ip.weight"]
                sd[f"to_v_ip.{i}.weight"] = s["ip_adapter"][f"{key_id}.to_v_ip.weight"]
            proc.load_state_dict(sd)
            attn.set_processor(proc.to(device=ref.device, dtype=ref.dtype))
            key_id += 2
        self.encoder_hid_proj = MultiIPAdapterImageProjection(projs).to(device=ref.device, dtype=ref.dtype)
        self.config["encoder_hid_dim_type"] = "ip_image_proj"
        return {}


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # number of transformer blocks per attention level; 0 = no attention at that level
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 0)
    mid_transformer_layers: int = 1
    # SD1.5's `attention_head_dim=8` is really the head COUNT (u_net_condition_modify.py:232-238)
    num_attention_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    time_embed_dim: int = 1280
    sample_size: int = 64
    extra: Dict[str, Any] = field(default_factory=dict)

    @staticmethod
    def sd15():
        return UNetConfig()

    @staticmethod
    def sdxl_base():
        """SDXL-base shape: 3 levels, depth (0,2,10), d = 64, ctx 2048."""
        return UNetConfig(block_out_channels=(320, 640, 1280), transformer_layers_per_block=(0, 2, 10),
                          mid_transformer_layers=10, num_attention_heads=(5, 10, 20), cross_attention_dim=2048,
                          use_linear_projection=True, sample_size=128)

    @staticmethod
    def tiny():
        """Same topology at toy width, for CPU-side structure tests and smoke()."""
        return UNetConfig(block_out_channels=(32, 64, 64, 64), num_attention_heads=(4, 4, 4, 4), cross_attention_dim=64,
                          norm_num_groups=8, time_embed_dim=128, sample_size=16)


def _derived(module, key, deps, build):
    """A tensor derived from parameters (concatenated / summed weights), cached on the module and rebuilt when any
    source parameter object or its in-place version changes (non-tensor entries of `deps`, e.g. a token count, compare by
    value).  Derived tensors are not parameters: state_dict keys stay the diffusers ones."""
    sig = tuple((id(t), t._version) if torch.is_tensor(t) else t for t in deps)
    hit = module.__dict__.get("_derived_" + key)
    if hit is None or hit[0] != sig:
        with torch.no_grad():
            hit = (sig, build())
        module.__dict__["_derived_" + key] = hit
    return hit[1]


def _tokens(x):
    """channels_last [B, C, h, w] -> token-major [B, h*w, C] (a view: the two layouts are the same bytes)"""
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, h * w, c)


def _image(t, h, w):
    """token-major [B, h*w, C] -> channels_last [B, C, h, w] (a view)"""
    b, _, c = t.shape
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2)


# GroupNorm statistics from the producer's epilogue (ops.conv3x3_gn / ops.linear_gn): asked for where the GroupNorm that follows
# is a two-launch one - the 64x64 / 32x32 levels at batch 1 (smaller tensors have a one-launch GroupNorm already, and their
# convolutions are split over K)
GN_FUSE_MIN_ROWS = 1024


def _with_gn(img, part):
    """attach a producer's GroupNorm partial sums to the tensor OBJECT handed on (GroupNormAct.forward looks for them)"""
    return ops.attach_gn_partials(img, part)


class Conv1x1(nn.Conv2d):
    """1x1 convolution run as a token-major GEMM (hipBLASLt) instead of a MIOpen convolution: same parameters and
    state_dict keys as nn.Conv2d(cin, cout, 1).  With channels-last activations the [B, h*w, C] token view is free,
    and MIOpen's 1x1 kernels at small spatial sizes accumulate with atomics (measured: run-to-run differences of 3e-2
    at 128->64 @ 2x2) while the GEMM is bit-reproducible."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1)

    def tokens(self, t, bias=None, residual=None, gn_groups=0):
        """gn_groups > 0: also try to get the GroupNorm partial sums of the result out of the GEMM's epilogue (for the
        GroupNorm that reads it next) -> (tokens, ops.GnPartials | None)"""
        b = self.bias if bias is None else bias
        if gn_groups and t.dim() == 3 and t.shape[0] * t.shape[1] >= GN_FUSE_MIN_ROWS:
            got = ops.linear_gn(t, self.weight.flatten(1), b, residual, t.shape[1], gn_groups)
            if got is not None:
                return got
        y = ops.linear(t, self.weight.flatten(1), b, residual=residual)
        return (y, None) if gn_groups else y

    def forward(self, x, residual=None):
        b, c, h, w = x.shape
        y = self.tokens(_tokens(x), residual=None if residual is None else _tokens(residual))
        return _image(y, h, w)


class GroupNormAct(nn.GroupNorm):
    """GroupNorm optionally fused with SiLU and with a per-(b, c) additive term (HIP kernel, channels-last)."""

    def __init__(self, groups, channels, eps, act=False):
        super().__init__(groups, channels, eps=eps, affine=True)
        self.act = act

    def forward(self, x, add=None):
        part = ops.gn_partials_of(x) if add is None and x.dim() == 4 else None
        if part is not None and part.groups == self.num_groups and part.C == x.shape[1] and part.B == x.shape[0] \
                and part.hw == x.shape[2] * x.shape[3] and x.is_contiguous(memory_format=torch.channels_last):
            # the kernel that wrote x also emitted its group sums: the whole GroupNorm is one launch
            return ops.groupnorm_apply_nhwc(x, part, self.num_groups, self.weight, self.bias, self.eps, self.act)
        return ops.groupnorm_silu_nhwc(x, self.num_groups, self.weight, self.bias, self.eps, self.act, add=add)

    def of_cat(self, x1, x2):
        """(norm(cat([x1, x2], 1)), cat([x1, x2], 1)): the concatenation is a by-product of the statistics pass"""
        if ops.groupnorm_cat_covers(x1, x2):
            return ops.groupnorm_silu_nhwc_cat(x1, x2, self.num_groups, self.weight, self.bias, self.eps, self.act)
        x = torch.cat([x1, x2], dim=1)
        return self.forward(x), x


class Attention(nn.Module):
    """The module the processors receive as `attn` (attributes listed in SURVEY.md 8b)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.inner_dim = inner
        self.is_cross_attention = cross_attention_dim is not None
        self.upcast_attention = False
        self.upcast_softmax = False
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0()
        self._proc_params = None
        # set by the pipeline for cross-attention: {"src": text tensor, "k": to_k(text), "v": to_v(text)} - the text
        # does not change during a generation, so its projections are computed once, not once per step
        self.kv_cache = None

    def qkv_weight(self):
        """[3C, C] concatenation of to_q / to_k / to_v for self-attention: one GEMM instead of three"""
        return _derived(self, "qkv", (self.to_q.weight, self.to_k.weight, self.to_v.weight),
                        lambda: torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight]).contiguous())

    def set_processor(self, processor):
        # a Module processor (AttnProcessor) registers as a child; drop that entry before a plain object replaces it
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor
        self._proc_params = None

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        """diffusers 0.27.2 `Attention.prepare_attention_mask` [recalled; un-vendored: parity unpinned]: a mask shorter or longer
        than the key length is zero-padded by target_length, then repeated per head: [B, 1|L, S] -> [B*heads, 1|L, S]
        (out_dim 3) or [B, heads, 1|L, S] (out_dim 4)."""
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3:
            if attention_mask.shape[0] < batch_size * self.heads:
                attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        elif out_dim == 4:
            attention_mask = attention_mask.unsqueeze(1).repeat_interleave(self.heads, dim=1)
        return attention_mask

    def head_to_batch_dim(self, t, out_dim=3):
        b, n, c = t.shape
        t = t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3)
        return t.reshape(b * self.heads, n, c // self.heads) if out_dim == 3 else t

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        return t.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(bh // self.heads, n, d * self.heads)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        # diffusers 0.27.2 Attention.forward drops kwargs the processor does not declare
        if self._proc_params is None:
            self._proc_params = set(inspect.signature(self.processor.__call__).parameters.keys())
        kw = {k: v for k, v in cross_attention_kwargs.items() if k in self._proc_params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return ops.linear(x, self.proj.weight, self.proj.bias, geglu=True)     # GEMM + bias + GEGLU in one kernel


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x, residual=None):
        h = self.net[0](x)
        return ops.linear(h, self.net[2].weight, self.net[2].bias, residual=residual)   # net[1] is Dropout(0)


class LNFold:
    """What BasicTransformerBlock hands to its attention processor when the LayerNorm in front of the attention is folded
    into the projection GEMM (dsc_linear_ln_f16): the norm module, the row statistics of the un-normalised stream (emitted
    by the GEMM that produced it) and the stream itself as the residual of to_out."""

    __slots__ = ("norm", "stats", "residual")

    def __init__(self, norm, stats, residual):
        self.norm, self.stats, self.residual = norm, stats, residual

    def folded(self, owner, key, weight, bias=None):
        """(W diag(gamma), W beta + b, row sums) of `weight` under self.norm, cached on `owner`"""
        n = self.norm
        deps = (weight, n.weight, n.bias) + (() if bias is None else (bias,))
        return _derived(owner, "lnfold_" + key, deps, lambda: ops.fold_layernorm(weight, bias, n.weight, n.bias))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def _can_fold(self, x, stats):
        """The three LayerNorms disappear into the neighbouring GEMMs when every one of those GEMMs runs on
        dsc_linear_ln_f16 and both attentions carry this package's stock processors (the private `_ln_fold` protocol)."""
        if stats is None or not x.is_cuda or x.dim() != 3:
            return False
        M, C = x.shape[0] * x.shape[1], x.shape[2]
        cover = ops.linear_kernel_covers
        if not (cover(M, 3 * C, C, x.dtype) and cover(M, C, C, x.dtype) and cover(M, 8 * C, C, x.dtype, True)):
            return False
        for a in (self.attn1, self.attn2):
            if type(a.processor) not in (AttnProcessor2_0, AttnProcessor) or a.to_q.bias is not None \
                    or type(a.to_q) is not nn.Linear or type(a.to_out[0]) is not nn.Linear \
                    or a.spatial_norm is not None or a.group_norm is not None or a.residual_connection \
                    or a.rescale_output_factor != 1.0 or a.inner_dim != C:
                return False
        return True

    def forward(self, x, encoder_hidden_states, cross_attention_kwargs, stats=None):
        kw = cross_attention_kwargs or {}
        if ops.USE_LN_FOLD and self._can_fold(x, stats):
            x, st = self.attn1(x, None, _ln_fold=LNFold(self.norm1, stats, x), **kw)
            x, st = self.attn2(x, encoder_hidden_states, _ln_fold=LNFold(self.norm2, st, x), **kw)
            f = LNFold(self.norm3, st, None)
            w2, b2, cvec = f.folded(self.ff, "geglu", self.ff.net[0].proj.weight, self.ff.net[0].proj.bias)
            h = ops.linear_ln(x, w2, b2, geglu=True, ln=(st, cvec, self.norm3.eps))
            return ops.linear(h, self.ff.net[2].weight, self.ff.net[2].bias, residual=x)
        # `x = attn(norm(x)) + x; h = next_norm(x)` pairs run as ONE add+LayerNorm launch (the same kwargs reach self-
        # and cross-attention, as in diffusers' BasicTransformerBlock)
        _, h = ops.add_layernorm(x, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x, h = ops.add_layernorm(x, self.attn1(h, None, **kw), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        a2 = self.attn2(h, encoder_hidden_states, **kw)
        if a2.shape[0] != x.shape[0]:                            # shared CFG prefix: the cross-attention output has the full batch
            x = x.repeat(a2.shape[0] // x.shape[0], 1, 1)
        x, h = ops.add_layernorm(x, a2, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self.ff(h, residual=x)


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, depth, cross_attention_dim, groups, use_linear_projection):
        super().__init__()
        self.use_linear_projection = use_linear_projection
        self.norm = GroupNormAct(groups, channels, 1e-6)
        if use_linear_projection:
            self.proj_in = nn.Linear(channels, channels)
            self.proj_out = nn.Linear(channels, channels)
        else:
            self.proj_in = Conv1x1(channels, channels)
            self.proj_out = Conv1x1(channels, channels)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, channels // heads, cross_attention_dim) for _ in range(depth)])

    def forward(self, x, encoder_hidden_states, cross_attention_kwargs):
        b, c, h, w = x.shape
        t = _tokens(self.norm(x))                               # channels-last: the token view is free
        w_in = self.proj_in.weight.flatten(1)
        st = None
        if ops.USE_LN_FOLD and t.is_cuda and ops.linear_kernel_covers(t.shape[0] * t.shape[1], c, c, t.dtype):
            t, st = ops.linear_ln(t, w_in, self.proj_in.bias, ln_stats=True)   # row statistics for the first block's norm1
        else:
            t = self.proj_in(t) if self.use_linear_projection else self.proj_in.tokens(t)
        for i, blk in enumerate(self.transformer_blocks):
            t = blk(t, encoder_hidden_states, cross_attention_kwargs, stats=st if i == 0 else None)
        res = _tokens(x)
        if res.shape[0] != t.shape[0] and not (res.is_cuda and ops.USE_RESIDUAL_WRAP and (res.shape[0] * res.shape[1]) % 128 == 0):
            # shared CFG prefix: x came in once per image.  (On the GPU the GEMM wraps a residual of fewer rows itself:
            # ops.linear / linear_gn, dsc_linear_f16.)
            res = res.repeat(t.shape[0] // res.shape[0], 1, 1)
        if self.use_linear_projection:
            t = ops.linear(t, self.proj_out.weight, self.proj_out.bias, residual=res)
            return _image(t, h, w)
        t, part = self.proj_out.tokens(t, residual=res, gn_groups=self.norm.num_groups)   # (+ the next GroupNorm's statistics)
        return _with_gn(_image(t, h, w), part)


def _conv3x3(x, weight, bias=None, residual=None):
    """3x3 / pad 1 convolution (+ bias) (+ residual): dsc_conv3x3_nhwc_f16 where it covers the shape (channels-last fp16,
    input channel counts that are multiples of 64), otherwise the library convolution (MIOpen) with
    the separate fused add."""
    if ops.conv3x3_supported(x, weight):
        return ops.conv3x3(x, weight, bias, residual)
    h = F.conv2d(x, weight, None, padding=1)
    if residual is not None:
        return ops.add_bias_residual(residual, h, bias)
    return h if bias is None else h + bias.view(1, -1, 1, 1)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = GroupNormAct(groups, cin, eps, act=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = GroupNormAct(groups, cout, eps, act=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv1x1(cin, cout) if cin != cout else None

    def temb_bias(self):
        """time_emb_proj.bias + conv1.bias: conv1 runs WITHOUT its bias and the sum rides in norm2's fused add"""
        return _derived(self, "tb", (self.time_emb_proj.bias, self.conv1.bias),
                        lambda: (self.time_emb_proj.bias + self.conv1.bias).contiguous())

    def forward(self, x, temb_act, temb_add=None):
        # conv biases never run as separate MIOpen bias kernels: conv1's is folded into the time-embedding term (which
        # is itself folded into norm2's load), conv2's into the shortcut GEMM's bias or the fused residual add
        if isinstance(x, tuple):            # up blocks: (hidden_states, skip) - norm1 produces their concatenation
            h, x = self.norm1.of_cat(*x)
        else:
            h = self.norm1(x)
        if temb_add is None:
            temb_add = F.linear(temb_act, self.time_emb_proj.weight, self.temb_bias())
        g = self.norm2.num_groups
        fuse = h.is_cuda and h.dtype == torch.float16 and h.shape[0] * h.shape[2] * h.shape[3] >= GN_FUSE_MIN_ROWS
        if fuse and ops.conv3x3_gn_rows(h, self.conv1.weight, g) > 0:
            # conv1 adds the time-embedding row itself (diffusers: hidden_states + temb between conv1 and norm2) and emits norm2's
            # statistics: norm2 is one launch
            h = self.norm2(ops.conv3x3_gn(h, self.conv1.weight, g, add=temb_add[:h.shape[0]]))
        else:
            h = self.norm2(_conv3x3(h, self.conv1.weight), add=temb_add)
        if self.conv_shortcut is None:
            if fuse and ops.conv3x3_gn_rows(h, self.conv2.weight, g) > 0:       # ... and conv2 the NEXT GroupNorm's (same grouping)
                return ops.conv3x3_gn(h, self.conv2.weight, g, bias=self.conv2.bias, residual=x)
            return _conv3x3(h, self.conv2.weight, self.conv2.bias, residual=x)
        h = _conv3x3(h, self.conv2.weight)
        b = _derived(self, "sb", (self.conv_shortcut.bias, self.conv2.bias),
                     lambda: (self.conv_shortcut.bias + self.conv2.bias).contiguous())
        bsz, _, hh, ww = h.shape
        t, part = self.conv_shortcut.tokens(_tokens(x), bias=b, residual=_tokens(h), gn_groups=g)
        return _with_gn(_image(t, hh, ww), part)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        if x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0 and ops.conv3x3_supported(x, self.conv.weight):
            return ops.conv3x3(x, self.conv.weight, self.conv.bias, stride2=True)      # even pixels of the stride-1 taps
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        if ops.conv3x3_supported(x, self.conv.weight, upsample=True):      # the upsampling happens in the halo gather
            return ops.conv3x3(x, self.conv.weight, self.conv.bias, upsample=True)
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    """resnets (+ attentions) (+ one down/upsampler); `attentions` is absent when depth == 0."""

    def __init__(self, res_io, temb_dim, cfg, heads, depth, down=False, up=False):
        super().__init__()
        cout = res_io[-1][1]
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, temb_dim, cfg.norm_num_groups, cfg.norm_eps) for i, o in res_io])
        self.has_attn = depth > 0
        if self.has_attn:
            self.attentions = nn.ModuleList([
                Transformer2DModel(cout, heads, depth, cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection)
                for _ in res_io])
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])


class UNetMidBlock(nn.Module):
    def __init__(self, c, temb_dim, cfg, heads, depth):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(c, heads, depth, cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection)]
            if depth > 0 else [])


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


class _Config(dict):
    __getattr__ = dict.get


class _EncoderHalf:
    """The part of the forward pass UNet2DConditionModel and ControlNetModel (modules/controlnet.py) share: time embedding,
    conv_in, down blocks, mid block.  Needs `cfg`, `conv_in`, `time_embedding`, `down_blocks`, `mid_block`, `_resnets()`."""

    def time_proj(self, timesteps, dim):
        """sinusoidal Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (reference :554); fractional t allowed."""
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
        ang = timesteps.float()[:, None] * freqs[None]
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)

    def _time_act(self, sample, timestep):
        """SiLU(time_embedding(Timesteps(t))) [B, time_embed_dim] - every ResnetBlock applies SiLU to temb first"""
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float32, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        te = self.time_embedding
        if sample.is_cuda and sample.dtype == torch.float16 and sample.shape[0] <= 8:
            # Timesteps -> linear_1 -> SiLU -> linear_2 -> SiLU as two few-row GEMV launches (13 kernels before)
            h1 = ops.linear_rows(timestep.float(), te.linear_1.weight, te.linear_1.bias, silu_out=True,
                                 sinusoid_dim=self.cfg.block_out_channels[0])
            return ops.linear_rows(h1, te.linear_2.weight, te.linear_2.bias, silu_out=True)
        emb = te(self.time_proj(timestep, self.cfg.block_out_channels[0]).to(sample.dtype))
        return F.silu(emb)

    def _few_channel_convs(self):
        return [c for c in (self.conv_in, getattr(self, "conv_out", None)) if c is not None]

    def _to_channels_last_once(self):
        if not self._channels_last:                              # conv weights to NHWC once: no per-call transposes
            self.to(memory_format=torch.channels_last)
            # MIOpen has no NHWC implicit-GEMM for 4 channels (it falls back to a 300-400 us naive kernel):
            # the 4-channel convolutions run NCHW and convert at the boundary when the hand-written kernels do not apply
            for conv in self._few_channel_convs():
                conv.weight.data = conv.weight.data.contiguous()
            self._channels_last = True

    def _conv_in(self, sample):
        ci = self.conv_in
        if (sample.is_cuda and sample.dtype == torch.float16 and ci.in_channels <= 16 and ci.out_channels % 8 == 0
                and ci.out_channels <= 512 and sample.shape[-1] % 8 == 0 and ops.USE_DSC_CONV):
            wt = _derived(self, "conv_in_t", (ci.weight,), lambda: ci.weight.reshape(ci.out_channels, -1).t().contiguous())
            return ops.conv3x3_fewcin(sample, wt, ci.bias, ci.out_channels)   # NCHW latents -> NHWC features, one launch
        return ci(sample.contiguous()).contiguous(memory_format=torch.channels_last)

    def _run_down(self, x, temb_act, tadd, encoder_hidden_states, cross_attention_kwargs, intrablock=None, skip0=None):
        """intrablock: T2I-Adapter features, one per down block (reference :1194-1230): added after the LAST resnet /
        attention pair of a cross-attention block (so the skip tensor carries it), after the whole block otherwise"""
        intrablock = None if intrablock is None else list(intrablock)
        skips = [x if skip0 is None else skip0]                      # (skip0: conv_in's output for the WHOLE batch when x is its shared half)
        for blk in self.down_blocks:
            extra = intrablock.pop(0) if intrablock else None
            last = len(blk.resnets) - 1
            for j, res in enumerate(blk.resnets):
                x = res(x, temb_act, tadd[res])
                if blk.has_attn:
                    x = blk.attentions[j](x, encoder_hidden_states, cross_attention_kwargs)
                    if x.shape[0] != skips[0].shape[0]:          # shared CFG prefix ended here: the skips so far, per row
                        skips = [torch.cat([s_] * (x.shape[0] // s_.shape[0])).contiguous(memory_format=torch.channels_last)
                                 for s_ in skips]
                    if j == last and extra is not None:
                        x = x + extra.to(x.dtype)
                skips.append(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
                skips.append(x)
            if not blk.has_attn and extra is not None:
                x = x + extra.to(x.dtype)
        return skips, x

    def _run_mid(self, x, temb_act, tadd, encoder_hidden_states, cross_attention_kwargs):
        x = self.mid_block.resnets[0](x, temb_act, tadd[self.mid_block.resnets[0]])
        for attn in self.mid_block.attentions:
            x = attn(x, encoder_hidden_states, cross_attention_kwargs)
        return self.mid_block.resnets[1](x, temb_act, tadd[self.mid_block.resnets[1]])

    def _temb_all(self, temb_act):
        """the per-ResNet `time_emb_proj(silu(temb))` GEMMs ([B,1280] x [cout,1280]; 22 in the SD1.5 UNet) as ONE GEMM over
        the concatenated weights: [B, sum of cout] (conv1's bias folded in, ResnetBlock2D.temb_bias)"""
        res = self._resnets()
        deps = tuple(p for r in res for p in (r.time_emb_proj.weight, r.time_emb_proj.bias, r.conv1.bias))
        W, Bv = _derived(self, "temb", deps, lambda: (torch.cat([r.time_emb_proj.weight for r in res]).contiguous(),
                                                       torch.cat([r.time_emb_proj.bias + r.conv1.bias for r in res])))
        if temb_act.is_cuda and temb_act.dtype == torch.float16 and temb_act.shape[0] <= 8:
            return ops.linear_rows(temb_act, W, Bv)
        return F.linear(temb_act, W, Bv)

    def _temb_views(self, allp):
        """{resnet: [B, cout] view of its columns}.  Offsets are multiples of 8 (16-byte aligned rows)."""
        out, off = {}, 0
        for r in self._resnets():
            n = r.time_emb_proj.out_features
            out[r] = allp[:, off:off + n]
            off += n
        return out

    def _all_temb_adds(self, temb_act):
        return self._temb_views(self._temb_all(temb_act))

    def temb_width(self):
        return sum(r.time_emb_proj.out_features for r in self._resnets())

    def temb_signature(self):
        """(id, in-place version) of every parameter `temb_add_table` reads - a caller that caches a table keys it by this, the
        way `_derived` keys derived weights, so an in-place weight update or a parameter swap invalidates the cache"""
        te = self.time_embedding
        deps = [te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias]
        for r in self._resnets():
            deps += [r.time_emb_proj.weight, r.time_emb_proj.bias, r.conv1.bias]
        return tuple((id(t), t._version) for t in deps)

    def temb_add_table(self, timesteps):
        """[len(timesteps), temb_width()]: row i = what `_temb_all` gives inside a forward at timestep i - for a caller that
        knows its timesteps up front (the fused denoising loop) and passes a row back as `forward(..., temb_adds=...)`.  The
        embedding depends on the timestep alone in this UNet (no class / added-condition embedding)."""
        ts = timesteps.reshape(-1).float()
        rows = []
        for i in range(0, ts.numel(), 8):                    # the few-row GEMV kernels take at most 8 rows
            chunk = ts[i:i + 8]
            probe = torch.empty((chunk.numel(), 1), device=ts.device, dtype=self.conv_in.weight.dtype)
            rows.append(self._temb_all(self._time_act(probe, chunk)))
        return torch.cat(rows, dim=0).contiguous()

    # ---- processor plumbing (reference :689-749)
    @property
    def attn_processors(self):
        return {f"{name}.processor": m.get_processor() for name, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor):
        attns = [(name, m) for name, m in self.named_modules() if isinstance(m, Attention)]
        if isinstance(processor, dict):
            if len(processor) != len(attns):
                raise ValueError(
                    f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                    f" number of attention layers: {len(attns)}. Please make sure to pass {len(attns)} processor classes.")
            for name, m in attns:
                m.set_processor(processor.pop(f"{name}.processor"))
        else:
            for _, m in attns:
                m.set_processor(processor)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device


class UNet2DConditionModel(_EncoderHalf, nn.Module, UNet2DConditionLoadersMixin_modify):
    def __init__(self, cfg: Optional[UNetConfig] = None):
        super().__init__()
        cfg = cfg or UNetConfig.sd15()
        self.cfg = cfg
        self.config = _Config(in_channels=cfg.in_channels, out_channels=cfg.out_channels, sample_size=cfg.sample_size,
                              cross_attention_dim=cfg.cross_attention_dim, block_out_channels=cfg.block_out_channels)
        ch = cfg.block_out_channels
        n = len(ch)
        td = cfg.time_embed_dim
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], td)
        self.down_blocks = nn.ModuleList()
        skip = [ch[0]]
        cin = ch[0]
        for i, c in enumerate(ch):
            io = [(cin if j == 0 else c, c) for j in range(cfg.layers_per_block)]
            self.down_blocks.append(_Block(io, td, cfg, cfg.num_attention_heads[i], cfg.transformer_layers_per_block[i],
                                           down=i < n - 1))
            skip += [c] * cfg.layers_per_block + ([c] if i < n - 1 else [])
            cin = c
        self.mid_block = UNetMidBlock(ch[-1], td, cfg, cfg.num_attention_heads[-1], cfg.mid_transformer_layers)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        rev_heads = list(reversed(cfg.num_attention_heads))
        rev_depth = list(reversed(cfg.transformer_layers_per_block))
        prev = ch[-1]
        for i, c in enumerate(rev):
            io = []
            for j in range(cfg.layers_per_block + 1):
                io.append(((prev if j == 0 else c) + skip.pop(), c))
            self.up_blocks.append(_Block(io, td, cfg, rev_heads[i], rev_depth[i], up=i < n - 1))
            prev = c
        assert not skip
        self._channels_last = False
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, ch[0], cfg.norm_eps, act=True)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def _first_block_repeats_shared_rows(self):
        """the shared CFG prefix hands the first cross-attention n query rows against 2n key rows: only this package's own
        processors (attention_modify._RegionProcessor) repeat the shared rows there - a user-installed processor gets the
        full batch instead"""
        from .attention_modify import _RegionProcessor
        blk = self.down_blocks[0].attentions[0].transformer_blocks[0]
        return isinstance(blk.attn1.processor, _RegionProcessor) and isinstance(blk.attn2.processor, _RegionProcessor)

    def _resnets(self):
        out = []
        for blk in self.down_blocks:
            out += list(blk.resnets)
        out += list(self.mid_block.resnets)
        for blk in self.up_blocks:
            out += list(blk.resnets)
        return out

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict=True, temb_adds=None,
                cfg_shared_prefix=False):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are not on the hot path (never passed by app.py)")
        # temb_adds: a [B, temb_width()] buffer holding this timestep's rows of temb_add_table() - the time-embedding path
        # is then not run (the captured step of the fused loop: its caller refreshes the buffer between replays)
        temb_act = None if temb_adds is not None else self._time_act(sample, timestep)
        self._to_channels_last_once()
        if getattr(self, "encoder_hid_proj", None) is not None and self.config.get("encoder_hid_dim_type") == "ip_image_proj":
            # reference :1030-1037 - the IP-Adapter image tokens travel with the text as a tuple
            if added_cond_kwargs is None or "image_embeds" not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `encoder_hid_dim_type` set to 'ip_image_proj' which "
                                 "requires the keyword argument `image_embeds` to be passed in  `added_conditions`")
            encoder_hidden_states = (encoder_hidden_states, self.encoder_hid_proj(added_cond_kwargs.get("image_embeds")))
        # cfg_shared_prefix (the caller's promise: rows [n:] of `sample` and of the time embedding equal rows [:n] - classifier-free
        # guidance's [x; x] input): everything before the first cross-attention is the same for both halves and runs on rows
        # [:n] only (conv_in, the first ResNet block, the first transformer block's GroupNorm / proj_in / self-attention / to_q);
        # the cross-attention processor repeats the shared rows against the full batch of text keys, and the batch is whole
        # again from there on (_run_down repeats the skip tensors collected so far)
        share = bool(cfg_shared_prefix) and sample.shape[0] % 2 == 0 and down_block_additional_residuals is None \
            and down_intrablock_additional_residuals is None and self.down_blocks[0].has_attn \
            and not isinstance(encoder_hidden_states, tuple) and self._first_block_repeats_shared_rows()
        # (conv_in itself runs on the whole batch: its output is the first skip tensor, which the up path wants per row - a second
        # image through the 4-channel convolution costs less than repeating the 320-channel result afterwards)
        x = self._conv_in(sample)
        skip0 = x
        if share:
            x = x[:sample.shape[0] // 2]
        tadd = self._temb_views(temb_adds) if temb_adds is not None else self._all_temb_adds(temb_act)
        if down_intrablock_additional_residuals is None and mid_block_additional_residual is None \
                and down_block_additional_residuals is not None:
            # legacy T2I-Adapter usage (reference :1200-1211): residuals without a mid residual are intra-block ones
            down_intrablock_additional_residuals, down_block_additional_residuals = down_block_additional_residuals, None
        skips, x = self._run_down(x, temb_act, tadd, encoder_hidden_states, cross_attention_kwargs,
                                  intrablock=down_intrablock_additional_residuals, skip0=skip0)
        if down_block_additional_residuals is not None:          # ControlNet hook (reference :1236-1245)
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
        x = self._run_mid(x, temb_act, tadd, encoder_hidden_states, cross_attention_kwargs)
        if mid_block_additional_residual is not None:            # reference :1269-1270
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                x = res((x, skips.pop()), temb_act, tadd[res])
                if blk.has_attn:
                    x = blk.attentions[j](x, encoder_hidden_states, cross_attention_kwargs)
            if hasattr(blk, "upsamplers"):
                x = blk.upsamplers[0](x)
        h = self.conv_norm_out(x)                                            # reference :1304-1307
        co = self.conv_out
        wcl = _derived(self, "conv_out_cl", (co.weight,), lambda: co.weight.contiguous(memory_format=torch.channels_last))
        if ops.conv3x3_supported(h, wcl):
            x = ops.conv3x3(h, wcl, co.bias, out_nchw=True)                   # the sampler's layout, no conversion pass
        else:
            x = co(h.contiguous()).contiguous()
        return UNet2DConditionOutput(sample=x) if return_dict else (x,)
