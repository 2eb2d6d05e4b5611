"""Oracle (test infrastructure only): region-biased cross-attention and the two attention processors.

Follows reference `source/modules/attention_modify.py`:
  * `region_attention`      <- `scaled_dot_product_attention_regionstate` (:74-103) with
                               `weight_func = lambda w, sigma, qk: w * sigma * qk.std()` (app.py:1004,
                               model_k_diffusion.py:967)
  * `attn_processor2_0`     <- `AttnProcessor2_0.__call__` (:414-503)
  * `attn_processor`        <- `AttnProcessor.__call__` (:106-207) + `get_attention_scores` (:39-70)
Pinned by tests/golden/attention_core.npz and tests/golden/processors.npz (tests/test_oracle_golden.py).
"""
import math

import torch


def default_weight_func(w, sigma, qk):
    """app.py:1004 - beta = sigma * std(a), std global and unbiased over the whole score tensor."""
    return w * sigma * qk.std()


def group_std(a, n_std_groups=1):
    """Unbiased std of the score tensor `a` [Bc,H,L,S] per std group; row b belongs to group b % n_std_groups.

    n_std_groups == 1 is the reference (`qk.std()` over everything, attention_modify.py:93-95).  The
    k-diffusion path only ever runs ONE image per call (external_k_diffusion.py:109-114 broadcasts c_in[B]
    against input[2B]), so its std group is one image's (uncond, cond) pair; micro-batching B images in the
    row layout [u_0..u_{B-1}, c_0..c_{B-1}] (model_k_diffusion.py:1021,1097) keeps that semantics with
    n_std_groups = B.  Returns a tensor of n_std_groups values.
    """
    Bc = a.shape[0]
    assert Bc % n_std_groups == 0
    g = a.reshape(Bc // n_std_groups, n_std_groups, -1).transpose(0, 1).reshape(n_std_groups, -1)
    return g.double().std(dim=1, unbiased=True)


def region_attention(q, k, v, w, sigma, scale=None, attn_mask=None, n_std_groups=1, fp16_rounding=False):
    """softmax(scale*q@k^T + mask + repeat_H(w * sigma * std(scale*q@k^T + mask))) @ v.

    q [Bc,H,L,d]; k,v [Bc,H,S,d]; w fp32 [Bw,L,S] with Bw | Bc*H (row bh of the flattened scores takes
    table row bh // (Bc*H/Bw), attention_modify.py:96-99).  `fp16_rounding` reproduces the points at which
    the reference's fp16 pipeline rounds (scores :90, std and sigma 0-dim fp16, in-place fp32 add into
    the fp16 scores :97, softmax :101, PV :103) on fp16-representable inputs.
    """
    Bc, H, L, d = q.shape
    S = k.shape[-2]
    scale = 1.0 / math.sqrt(d) if scale is None else scale          # :77 (attn.scale is ignored)
    f = (lambda t: t.half().float()) if fp16_rounding else (lambda t: t)
    q, k, v = q.float(), k.float(), v.float()
    a = f(f(q @ k.transpose(-2, -1)) * scale)                       # :90
    if attn_mask is not None:
        a = f(a + attn_mask.float())                                # :85-91
    std = group_std(a, n_std_groups)                                # weight_func, :95
    if fp16_rounding:
        std = std.half()
        sigma = torch.as_tensor(sigma).half()
    std = std.float()
    sigma = float(sigma)
    Bw = w.shape[0]
    rep = (Bc * H) // Bw                                            # :96
    wrow = torch.repeat_interleave(w.float(), rep, dim=0).reshape(Bc, H, L, S)
    gidx = torch.arange(Bc) % n_std_groups
    bias = (wrow * sigma) * std[gidx].reshape(Bc, 1, 1, 1)          # w * sigma * std, fp32 (app.py:1004)
    a = f(a + bias)                                                 # :97
    p = f(torch.softmax(a, dim=-1))                                 # :101 (dropout p=0, :102)
    return f(p @ v)                                                 # :103


def _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain):
    residual = hidden_states                                        # :425
    img_sequence_length = hidden_states.shape[1]                    # :427 (dim 1 also for 4-D input)
    input_ndim = hidden_states.ndim
    if input_ndim == 4:                                             # :433-435
        b, c, hh, ww = hidden_states.shape
        hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
    is_xattn = encoder_hidden_states is not None and region_prompt is not None   # :438
    query = attn.to_q(hidden_states)                                # :458
    enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states
    key, value = attn.to_k(enc), attn.to_v(enc)                     # :465-466
    B = hidden_states.shape[0]
    H = attn.heads
    d = key.shape[-1] // H
    qh = query.view(B, -1, H, d).transpose(1, 2)                    # :471-474
    kh = key.view(B, -1, H, d).transpose(1, 2)
    vh = value.view(B, -1, H, d).transpose(1, 2)
    if is_xattn and isinstance(region_prompt["region_state"], dict):           # :479
        w = region_prompt["region_state"][img_sequence_length]     # KeyError if L not in the table (:481)
        o = core_region(qh, kh, vh, w, region_prompt["sigma"], region_prompt["weight_func"])
    else:
        o = core_plain(qh, kh, vh)
    o = o.transpose(1, 2).reshape(B, -1, H * d)                     # :487
    o = attn.to_out[1](attn.to_out[0](o))                           # :491-493
    if input_ndim == 4:
        o = o.transpose(-1, -2).reshape(b, c, hh, ww)               # :495-496
    if attn.residual_connection:
        o = o + residual                                            # :498-499
    return o / attn.rescale_output_factor                           # :501


def attn_processor2_0(attn, hidden_states, encoder_hidden_states=None, region_prompt=None, n_std_groups=1):
    """AttnProcessor2_0.__call__ (:414-503); scale = 1/sqrt(d) on both branches (:77, SDPA default)."""
    def core_region(q, k, v, w, sigma, weight_func):
        d = q.shape[-1]
        a = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d))
        Bc, H, L, S = a.shape
        flat = a.reshape(-1, L, S)                                  # :94
        cw = weight_func(w, sigma, flat)                            # :95
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v

    def core_plain(q, k, v):
        d = q.shape[-1]
        return torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d)), dim=-1) @ v   # :483-485

    return _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain)


def attn_processor(attn, hidden_states, encoder_hidden_states=None, region_prompt=None):
    """AttnProcessor.__call__ (:106-207): same math through baddbmm/bmm with alpha = attn.scale (:57-63)."""
    def core_region(q, k, v, w, sigma, weight_func):
        a = (q @ k.transpose(-2, -1)) * attn.scale                  # get_attention_scores :57-63
        Bc, H, L, S = a.shape
        flat = a.reshape(-1, L, S)
        cw = weight_func(w, sigma, flat)                            # :167
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v            # :173-175

    def core_plain(q, k, v):
        return torch.softmax((q @ k.transpose(-2, -1)) * attn.scale, dim=-1) @ v   # :187-188

    return _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain)
