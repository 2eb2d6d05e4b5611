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


def _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain, ip_branch=None,
                 attention_mask=None):
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
    mk = {} if attention_mask is None else {"mask": attention_mask}   # the processors that take one pass it on (:144, :448-452)
    if is_xattn and isinstance(region_prompt["region_state"], dict):           # :479
        w = region_prompt["region_state"][img_sequence_length]     # KeyError if L not in the table (:481)
        o = core_region(qh, kh, vh, w, region_prompt["sigma"], region_prompt["weight_func"], **mk)
    else:
        o = core_plain(qh, kh, vh, **mk)
    o = o.transpose(1, 2).reshape(B, -1, H * d)                     # :487
    if ip_branch is not None:
        o = ip_branch(o, qh, B, H, d)                               # :649-683 / :354-384: before the out projection
    o = attn.to_out[1](attn.to_out[0](o))                           # :491-493
    if input_ndim == 4:
        o = o.transpose(-1, -2).reshape(b, c, hh, ww)               # :495-496
    if attn.residual_connection:
        o = o + residual                                            # :498-499
    return o / attn.rescale_output_factor                           # :501


def attn_processor2_0(attn, hidden_states, encoder_hidden_states=None, region_prompt=None, n_std_groups=1, attention_mask=None):
    """AttnProcessor2_0.__call__ (:414-503); scale = 1/sqrt(d) on both branches (:77, SDPA default).  attention_mask: what
    `attn.prepare_attention_mask` returned, [B*H, 1|L, S]; viewed [B, H, ., S] (:448-452).  With a region table that 4-D
    mask meets the in-place `attn_bias += attn_mask` into an [L, S] tensor (:89), which torch refuses: RuntimeError, as in
    the reference (tests/golden/attention_masks.npz p2/raises)."""
    def core_region(q, k, v, w, sigma, weight_func, mask=None):
        d = q.shape[-1]
        a = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d))
        Bc, H, L, S = a.shape
        if mask is not None:
            bias = torch.zeros(L, S, dtype=q.dtype)
            bias += mask.view(Bc, H, -1, S)                         # :89 - raises for any 4-D mask
            a = a + bias
        flat = a.reshape(-1, L, S)                                  # :94
        cw = weight_func(w, sigma, flat)                            # :95
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v

    def core_plain(q, k, v, mask=None):
        d = q.shape[-1]
        a = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d))
        if mask is not None:
            a = a + mask.view(a.shape[0], a.shape[1], -1, a.shape[-1])
        return torch.softmax(a, dim=-1) @ v                         # :483-485

    return _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain,
                        attention_mask=attention_mask)


def attn_processor(attn, hidden_states, encoder_hidden_states=None, region_prompt=None, attention_mask=None):
    """AttnProcessor.__call__ (:106-207): same math through baddbmm/bmm with alpha = attn.scale (:57-63); the attention mask
    ([B*H, 1|L, S]) is baddbmm's additive input (beta = 1, :52-63), so weight_func's std sees the masked scores (:166-167)."""
    def core_region(q, k, v, w, sigma, weight_func, mask=None):
        a = (q @ k.transpose(-2, -1)) * attn.scale                  # get_attention_scores :57-63
        Bc, H, L, S = a.shape
        flat = a.reshape(-1, L, S)
        if mask is not None:
            flat = flat + mask
        cw = weight_func(w, sigma, flat)                            # :167
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v            # :173-175

    def core_plain(q, k, v, mask=None):
        a = (q @ k.transpose(-2, -1)) * attn.scale
        if mask is not None:
            a = (a.reshape(-1, a.shape[-2], a.shape[-1]) + mask).reshape(a.shape)
        return torch.softmax(a, dim=-1) @ v                         # :187-188

    return _proc_common(attn, hidden_states, encoder_hidden_states, region_prompt, core_region, core_plain,
                        attention_mask=attention_mask)


# ----------------------------------------------------------------------------- IP-Adapter processors
def split_ip_hidden_states(encoder_hidden_states, num_tokens):
    """attention_modify.py:560-577 / :268-282: `(text, [image tokens per adapter])`, or the deprecated single tensor whose
    last num_tokens[0] rows are the (single) adapter's image tokens."""
    if isinstance(encoder_hidden_states, tuple):
        return encoder_hidden_states
    end_pos = encoder_hidden_states.shape[1] - num_tokens[0]
    return encoder_hidden_states[:, :end_pos, :], [encoder_hidden_states[:, end_pos:, :]]


def ip_mask_downsample(mask, batch_size, num_queries, value_embed_dim):
    """diffusers 0.27.2 `IPAdapterMaskProcessor.downsample` (called at attention_modify.py:672-675, :373-376).  The
    package is absent from /root/reference and from this image: restated from the published algorithm - PARITY UNPINNED.
    mask [1, h, w] -> [batch_size, num_queries, value_embed_dim]: bicubic resize to the query grid that keeps the
    mask's aspect ratio, flatten, pad / truncate to num_queries, broadcast over the channels."""
    o_h, o_w = mask.shape[1], mask.shape[2]
    ratio = o_w / o_h
    mask_h = int(math.sqrt(num_queries / ratio))
    mask_h = int(mask_h) + int((num_queries % int(mask_h)) != 0)
    mask_w = num_queries // mask_h
    m = torch.nn.functional.interpolate(mask.unsqueeze(0), size=(mask_h, mask_w), mode="bicubic").squeeze(0)
    if m.shape[0] < batch_size:
        m = m.repeat(batch_size, 1, 1)
    m = m.view(m.shape[0], -1)
    area = mask_h * mask_w
    if area < num_queries:
        m = torch.nn.functional.pad(m, (0, num_queries - m.shape[1]), value=0.0)
    if area > num_queries:
        m = m[:, :num_queries]
    return m.view(m.shape[0], m.shape[1], 1).repeat(1, 1, value_embed_dim)


def _ip_branch(ip, attn, ip_hidden_states, ip_adapter_masks, scale_of):
    """:636-683 / :341-384 - validation of the masks, then per adapter: K/V projections of the image tokens, plain
    softmax attention of the SAME queries over them, optional mask multiply, `hidden += scale * ip_out`."""
    if ip_adapter_masks is not None:
        if not isinstance(ip_adapter_masks, torch.Tensor) or ip_adapter_masks.ndim != 4:
            raise ValueError("ip_adapter_mask should be a tensor with shape [num_ip_adapter, 1, height, width].")
        if len(ip_adapter_masks) != len(ip.scale):
            raise ValueError("Number of ip_adapter_masks must match number of IP-Adapters")
    else:
        ip_adapter_masks = [None] * len(ip.scale)

    def branch(o, qh, B, H, d):
        for cur, sc, to_k_ip, to_v_ip, mask in zip(ip_hidden_states, ip.scale, ip.to_k_ip, ip.to_v_ip, ip_adapter_masks):
            k = to_k_ip(cur).view(B, -1, H, d).transpose(1, 2)
            v = to_v_ip(cur).view(B, -1, H, d).transpose(1, 2)
            a = torch.softmax((qh @ k.transpose(-2, -1)) * scale_of(d), dim=-1) @ v      # :661-663 SDPA / :362-363
            a = a.transpose(1, 2).reshape(B, -1, H * d)
            if mask is not None:
                a = a * ip_mask_downsample(mask, B, a.shape[1], a.shape[2]).to(a.dtype)
            o = o + sc * a                                                               # :683 / :384
        return o
    return branch


def ip_adapter_attn_processor2_0(ip, attn, hidden_states, encoder_hidden_states=None, region_prompt=None,
                                 ip_adapter_masks=None):
    """IPAdapterAttnProcessor2_0.__call__ (:547-700).  `ip` carries num_tokens, scale (list), to_k_ip, to_v_ip."""
    text, ip_hidden_states = split_ip_hidden_states(encoder_hidden_states, ip.num_tokens)
    branch = _ip_branch(ip, attn, ip_hidden_states, ip_adapter_masks, lambda d: 1.0 / math.sqrt(d))

    def core_region(q, k, v, w, sigma, weight_func):
        d = q.shape[-1]
        a = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d))
        Bc, H, L, S = a.shape
        flat = a.reshape(-1, L, S)
        cw = weight_func(w, sigma, flat)
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v

    def core_plain(q, k, v):
        return torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(q.shape[-1])), dim=-1) @ v

    return _proc_common(attn, hidden_states, text, region_prompt, core_region, core_plain, ip_branch=branch)


def ip_adapter_attn_processor(ip, attn, hidden_states, encoder_hidden_states=None, region_prompt=None,
                              ip_adapter_masks=None):
    """IPAdapterAttnProcessor.__call__ (:247-404): the same through baddbmm/bmm with alpha = attn.scale."""
    text, ip_hidden_states = split_ip_hidden_states(encoder_hidden_states, ip.num_tokens)
    branch = _ip_branch(ip, attn, ip_hidden_states, ip_adapter_masks, lambda d: attn.scale)

    def core_region(q, k, v, w, sigma, weight_func):
        a = (q @ k.transpose(-2, -1)) * attn.scale
        Bc, H, L, S = a.shape
        flat = a.reshape(-1, L, S)
        cw = weight_func(w, sigma, flat)
        flat = flat + torch.repeat_interleave(cw, flat.shape[0] // cw.shape[0], dim=0)
        return torch.softmax(flat.reshape(Bc, H, L, S), dim=-1) @ v

    def core_plain(q, k, v):
        return torch.softmax((q @ k.transpose(-2, -1)) * attn.scale, dim=-1) @ v

    return _proc_common(attn, hidden_states, text, region_prompt, core_region, core_plain, ip_branch=branch)
