"""Region-map encoder (host side, once per generation) - counterpart of reference
`source/modules/encode_region_map_function.py` (`encode_region_map_sp` :21-77, `encode_region_map` :79-124).

Same signatures, same output contract `{L: FloatTensor[Bc*n, L, S]}` on the CPU in fp32, same quirks (SURVEY.md 8a
q1-q4: negative ids are overwritten by the positive ids; `state=None` still yields a dict of zeros; an empty region
turns the whole level positive because `== max` with max 0 is all-true; batch order [u, c, u, c, ...]).

OpenCV is not available on the target box, so `cv2.resize(mask, (w_r, h_r), INTER_CUBIC)` (:50) is replaced by
`_resize_cubic_u8`, a vectorised separable Keys bicubic (a = -0.75, half-pixel centres, replicated borders,
round-half-up, saturate to uint8) following OpenCV's published algorithm.  For masks that are constant on aligned
blocks (what the goldens pin) every kernel gives the block value; for arbitrary masks the cv2 bit pattern is
**parity unpinned**.
"""
import math

import numpy as np
import torch

_A = -0.75


def _taps(n_dst, n_src):
    """4 source indices and Keys weights per destination index"""
    f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
    base = np.floor(f)
    x = f - base
    w = np.stack([((_A * (x + 1) - 5 * _A) * (x + 1) + 8 * _A) * (x + 1) - 4 * _A,
                  ((_A + 2) * x - (_A + 3)) * x * x + 1,
                  ((_A + 2) * (1 - x) - (_A + 3)) * (1 - x) * (1 - x) + 1], axis=1)
    w = np.concatenate([w, 1.0 - w.sum(axis=1, keepdims=True)], axis=1)
    idx = np.clip(base[:, None].astype(np.int64) + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, w


def _resize_cubic_u8(img, dsize):
    w_r, h_r = dsize
    src = np.asarray(img, dtype=np.float64)
    xi, xw = _taps(w_r, src.shape[1])
    yi, yw = _taps(h_r, src.shape[0])
    tmp = np.einsum("hwt,wt->hw", src[:, xi], xw)
    out = np.einsum("htw,ht->hw", tmp[yi, :], yw)
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def _match_starts(ids, toks):
    """every index i with ids[i : i+len(toks)] == toks (the window compare of :60-61,66-67)"""
    n = len(toks)
    if n == 0 or ids is None:
        return []
    return [i for i in range(len(ids)) if ids[i:i + n] == toks]


def _level_column(v, w_r, h_r):
    """one region's per-token value over the flattened level: +weight inside, -mask_outsides outside (:49-53)"""
    m = _resize_cubic_u8(np.array(v["map"] < 255, dtype=np.uint8), (w_r, h_r))
    inside = (m == m.max())
    col = inside.astype(np.float64) * float(v["weight"])
    col[col == 0] = -1.0 * float(v["mask_outsides"])
    return col.reshape(-1)


def encode_region_map_sp(state, tokenizer, unet, width, height, scale_ratio=8, text_ids=None,
                         do_classifier_free_guidance=True):
    if text_ids is None:
        return torch.FloatTensor(0)
    as_list = lambda a: a.reshape(-1).tolist() if isinstance(a, (np.ndarray, torch.Tensor)) else None  # noqa: E731
    uncond, cond = as_list(text_ids[0]), as_list(text_ids[1])
    tables = {}
    for _ in unet.down_blocks:                                   # one level per down block (:29)
        w_r, h_r = int(math.ceil(width / scale_ratio)), int(math.ceil(height / scale_ratio))
        L, S = w_r * h_r, len(cond)
        t_cond = np.zeros((L, S), dtype=np.float32)
        t_uncond = np.zeros((L, S), dtype=np.float32)
        for phrase, v in (state.items() if state is not None else ()):
            if v["map"] is None:
                continue
            toks = tokenizer(phrase, max_length=tokenizer.model_max_length, truncation=True,
                             add_special_tokens=False).input_ids
            col = _level_column(v, w_r, h_r).astype(np.float32)
            hits = 0
            for ids, table in ((cond, t_cond), (uncond, t_uncond)):
                for i in _match_starts(ids, toks):
                    hits += 1
                    table[:, i:i + len(toks)] += col[:, None]
            if hits == 0:
                print(f"tokens {toks} not found in text")
        pair = [t_uncond, t_cond] if do_classifier_free_guidance else [t_cond]
        tables[L] = torch.from_numpy(np.stack(pair))
        scale_ratio *= 2
    return tables


def encode_region_map(pipe, state, width, height, num_images_per_prompt, text_ids=None):
    negative_ids, prompt_ids = text_ids[0], text_ids[1]
    if prompt_ids is None:
        return torch.FloatTensor(0)
    prompt_ids = np.array(prompt_ids)
    # quirk q1 (:91): the negative ids are REPLACED by the positive ids, so uncond rows get the cond table
    negative_ids = np.array(prompt_ids) if negative_ids is not None else None
    n_prompt = prompt_ids.shape[0]
    pos = np.split(prompt_ids, n_prompt)
    neg = np.split(negative_ids, n_prompt) if negative_ids is not None else [None] * n_prompt
    if not isinstance(state, list):
        state = [state]
    if len(state) < n_prompt:                                    # (:100-101) nests the list, as the reference does
        state = [state] + [None] * int(n_prompt - len(state))
    merged = {}
    for i in range(n_prompt):
        level_tables = encode_region_map_sp(state[i], pipe.tokenizer, pipe.unet, width, height,
                                            scale_ratio=pipe.vae_scale_factor, text_ids=[neg[i], pos[i]],
                                            do_classifier_free_guidance=pipe.do_classifier_free_guidance)
        for L, t in level_tables.items():
            merged[L] = torch.cat((merged[L], t)) if L in merged else t
    return {L: t.repeat(num_images_per_prompt, 1, 1) for L, t in merged.items()}
