"""Oracle (test infrastructure only): region-map encoder.

Follows reference `source/modules/encode_region_map_function.py`:
  * `encode_region_map_sp` (:21-77)  - per UNet level, resize the user mask, binarise at its max, scale,
    scatter into every token span matching the phrase's ids
  * `encode_region_map`    (:79-124) - per prompt, concat, repeat per image
Pinned by tests/golden/region_encoder.npz for masks constant on 64-px-aligned blocks (incl. quirks q1-q4 of
SURVEY.md 8a).  `resize_cubic_u8` restates OpenCV's `cv2.resize(..., INTER_CUBIC)` uint8 path from its
published algorithm (a = -0.75, half-pixel centres, replicated border, round-half-up, saturate); OpenCV is not
installed here, so for masks that are NOT block-aligned this kernel is **parity unpinned**.
"""
import math

import numpy as np
import torch


def _cubic_coeffs(x, A=-0.75):
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    return np.array([c0, c1, c2, 1.0 - c0 - c1 - c2])


def resize_cubic_u8(img, dsize):
    """uint8 [H,W] -> uint8 [h_r,w_r]; dsize = (w_r, h_r) as in cv2.resize (encode_region_map_function.py:50)."""
    w_r, h_r = dsize
    H, W = img.shape
    src = img.astype(np.float64)

    def taps(n_dst, n_src):
        idx = np.zeros((n_dst, 4), dtype=np.int64)
        cf = np.zeros((n_dst, 4))
        sc = n_src / n_dst
        for i in range(n_dst):
            f = (i + 0.5) * sc - 0.5
            s = math.floor(f)
            cf[i] = _cubic_coeffs(f - s)
            idx[i] = np.clip(np.arange(s - 1, s + 3), 0, n_src - 1)
        return idx, cf

    yi, yc = taps(h_r, H)
    xi, xc = taps(w_r, W)
    tmp = np.zeros((H, w_r))
    for t in range(4):
        tmp += src[:, xi[:, t]] * xc[None, :, t]
    out = np.zeros((h_r, w_r))
    for t in range(4):
        out += tmp[yi[:, t], :] * yc[:, t, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def encode_region_map_sp(state, tokenizer, n_levels, width, height, scale_ratio=8, text_ids=None,
                         do_classifier_free_guidance=True):
    """encode_region_map_function.py:21-77; `n_levels` = len(unet.down_blocks) (:29)."""
    if text_ids is None:                                                      # :22-23
        return torch.FloatTensor(0)
    uncond, cond = text_ids[0], text_ids[1]
    tolist = lambda a: a.reshape(-1).tolist() if isinstance(a, (np.ndarray, torch.Tensor)) else None  # noqa: E731
    cond, uncond = tolist(cond), tolist(uncond)                               # :27-28
    w_tensors = {}
    for _ in range(n_levels):
        c = len(cond)
        w_r, h_r = int(math.ceil(width / scale_ratio)), int(math.ceil(height / scale_ratio))   # :31
        ret_cond = torch.zeros((1, w_r * h_r, c), dtype=torch.float32)
        ret_uncond = torch.zeros((1, w_r * h_r, c), dtype=torch.float32)
        if state is not None:                                                 # :36 (None still yields zeros: q2)
            for phrase, v in state.items():
                if v["map"] is None:
                    continue
                toks = tokenizer(phrase, max_length=tokenizer.model_max_length, truncation=True,
                                 add_special_tokens=False).input_ids          # :42-47
                m = np.array(v["map"] < 255, dtype=np.uint8)                  # :49
                m = resize_cubic_u8(m, (w_r, h_r))                            # :50
                m = (m == np.max(m)).astype(float)                            # :51 (max == 0 -> all true: q3)
                m = m * float(v["weight"])                                    # :52
                m[m == 0] = -1 * float(v["mask_outsides"])                    # :53
                ret = torch.from_numpy(m).reshape(-1, 1).repeat(1, len(toks)) # :54-57
                n = len(toks)
                found = False
                for ids, dst in ((cond, ret_cond), (uncond, ret_uncond)):     # :59-69
                    if ids is None:
                        continue
                    for i in range(len(ids)):
                        if ids[i:i + n] == toks:
                            found = True
                            dst[0, :, i:i + n] += ret
                if not found:
                    print(f"tokens {toks} not found in text")                 # :71-72
        w_tensors[w_r * h_r] = torch.cat([ret_uncond, ret_cond]) if do_classifier_free_guidance else ret_cond  # :74
        scale_ratio *= 2                                                      # :75
    return w_tensors


def encode_region_map(tokenizer, n_levels, vae_scale_factor, do_classifier_free_guidance, state, width, height,
                      num_images_per_prompt, text_ids=None):
    """encode_region_map_function.py:79-124 with the `pipe` attributes it reads passed explicitly."""
    neg_ids, pos_ids = text_ids[0], text_ids[1]
    if pos_ids is None:                                                       # :88-89
        return torch.FloatTensor(0)
    pos_ids = np.array(pos_ids)
    neg_ids = np.array(pos_ids) if neg_ids is not None else None              # :91 - negative ids := positive ids (q1)
    n_prompt = pos_ids.shape[0]
    pos_l = np.split(pos_ids, n_prompt)
    neg_l = np.split(neg_ids, n_prompt) if neg_ids is not None else None
    if not isinstance(state, list):
        state = [state]
    if len(state) < n_prompt:                                                 # :100-101 (nests the list: reference quirk)
        state = [state] + [None] * int(n_prompt - len(state))
    per_prompt = []
    for i in range(n_prompt):
        ids = [neg_l[i], pos_l[i]] if neg_l is not None else [None, pos_l[i]]
        per_prompt.append(encode_region_map_sp(state[i], tokenizer, n_levels, width, height,
                                               scale_ratio=vae_scale_factor, text_ids=ids,
                                               do_classifier_free_guidance=do_classifier_free_guidance))
    merged = {}
    for d in per_prompt:                                                      # :107-115
        for key, t in d.items():
            merged[key] = torch.cat((merged[key], t)) if key in merged else t
    return {key: t.repeat(num_images_per_prompt, 1, 1) for key, t in merged.items()}   # :118-122
