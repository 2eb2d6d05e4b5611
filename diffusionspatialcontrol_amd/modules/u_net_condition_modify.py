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

Device work: convolutions and projection GEMMs go to MIOpen / hipBLASLt through torch (plain library
GEMMs); GroupNorm+SiLU, self-attention, region cross-attention run in libdsc_hip.so (see ..ops).
"""
import inspect
import math
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .attention_modify import AttnProcessor2_0


class UNet2DConditionLoadersMixin_modify:
    """Defined nowhere in the reference tree although imported at u_net_condition_modify.py:23; the IP-Adapter
    loader calls `_load_ip_adapter_weights` on it (ip_adapter.py:231).  IP-Adapter is outside the hot path
    (SURVEY.md 8f rank 2): the hook exists so the import surface resolves, and says so when used."""

    def _load_ip_adapter_weights(self, state_dicts, low_cpu_mem_usage=False):
        raise NotImplementedError("IP-Adapter weights are outside the MI355X hot path built so far")


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


class Conv1x1(nn.Conv2d):
    """1x1 convolution run as a GEMM over channels (hipBLASLt) instead of a MIOpen convolution: same parameters and
    state_dict keys as nn.Conv2d(cin, cout, 1).  MIOpen's 1x1 kernels at small spatial sizes accumulate with atomics
    (measured: run-to-run differences of 3e-2 at 128->64 @ 2x2), the GEMM is bit-reproducible."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1)

    def forward(self, x, residual=None):
        b, c, h, w = x.shape
        w2 = self.weight.view(self.out_channels, c)
        y = torch.matmul(w2, x.reshape(b, c, h * w))
        y = y + self.bias.view(1, -1, 1)
        y = y.view(b, self.out_channels, h, w)
        return y if residual is None else y + residual

    def tokens(self, x):
        """[B, C, h, w] -> projected tokens [B, h*w, Cout] without an NCHW->NLC copy"""
        b, c, h, w = x.shape
        return F.linear(x.reshape(b, c, h * w).transpose(1, 2), self.weight.view(self.out_channels, c), self.bias)

    def from_tokens(self, t, h, w, residual):
        """tokens [B, L, C] -> [B, Cout, h, w] (+ residual)"""
        b, L, c = t.shape
        y = torch.matmul(self.weight.view(self.out_channels, c), t.transpose(1, 2)) + self.bias.view(1, -1, 1)
        return y.view(b, self.out_channels, h, w) + residual


class GroupNormAct(nn.GroupNorm):
    """GroupNorm optionally fused with SiLU (HIP kernel on the GPU)."""

    def __init__(self, groups, channels, eps, act=False):
        super().__init__(groups, channels, eps=eps, affine=True)
        self.act = act

    def forward(self, x):
        return ops.groupnorm_silu(x, self.num_groups, self.weight, self.bias, self.eps, self.act)


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

    def set_processor(self, processor):
        # a Module processor (AttnProcessor) registers as a child; drop that entry before a plain object replaces it
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor
        self._proc_params = None

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return None
        raise NotImplementedError("additive attention masks are not on the hot path (never passed by app.py)")

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
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states, cross_attention_kwargs):
        kw = cross_attention_kwargs or {}
        x = self.attn1(self.norm1(x), None, **kw) + x          # the same kwargs reach self- and cross-attention
        x = self.attn2(self.norm2(x), encoder_hidden_states, **kw) + x
        return self.ff(self.norm3(x)) + x


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
        res = x
        x = self.norm(x)
        if self.use_linear_projection:
            x = self.proj_in(x.permute(0, 2, 3, 1).reshape(b, h * w, c))
        else:
            x = self.proj_in.tokens(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states, cross_attention_kwargs)
        if self.use_linear_projection:
            return self.proj_out(x).reshape(b, h, w, c).permute(0, 3, 1, 2) + res
        return self.proj_out.from_tokens(x, h, w, res)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = GroupNormAct(groups, cin, eps, act=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = GroupNormAct(groups, cout, eps, act=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv1x1(cin, cout) if cin != cout else None

    def forward(self, x, temb_act):
        h = self.conv1(self.norm1(x))
        h = h + self.time_emb_proj(temb_act)[:, :, None, None]
        h = self.conv2(self.norm2(h))
        return self.conv_shortcut(x, residual=h) if self.conv_shortcut is not None else x + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
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


class UNet2DConditionModel(nn.Module, UNet2DConditionLoadersMixin_modify):
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
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, ch[0], cfg.norm_eps, act=True)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

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

    def time_proj(self, timesteps, dim):
        """sinusoidal Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (reference :554); fractional t allowed."""
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
        ang = timesteps.float()[:, None] * freqs[None]
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict=True):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are not on the hot path (never passed by app.py)")
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float32, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timestep, self.cfg.block_out_channels[0]).to(sample.dtype))
        temb_act = F.silu(emb)                                   # every ResnetBlock applies SiLU to temb first
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                x = res(x, temb_act)
                if blk.has_attn:
                    x = blk.attentions[j](x, encoder_hidden_states, cross_attention_kwargs)
                skips.append(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
                skips.append(x)
        if down_block_additional_residuals is not None:          # ControlNet hook (reference :1236-1245)
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
        x = self.mid_block.resnets[0](x, temb_act)
        for attn in self.mid_block.attentions:
            x = attn(x, encoder_hidden_states, cross_attention_kwargs)
        x = self.mid_block.resnets[1](x, temb_act)
        if mid_block_additional_residual is not None:            # reference :1269-1270
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                x = res(torch.cat([x, skips.pop()], dim=1), temb_act)
                if blk.has_attn:
                    x = blk.attentions[j](x, encoder_hidden_states, cross_attention_kwargs)
            if hasattr(blk, "upsamplers"):
                x = blk.upsamplers[0](x)
        x = self.conv_out(self.conv_norm_out(x))                 # reference :1304-1307
        return UNet2DConditionOutput(sample=x) if return_dict else (x,)
