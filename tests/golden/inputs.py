"""Seeded input recipes shared by tests/golden/make_golden.py (build container, runs the reference) and
by the parity tests (any box).  numpy PCG64 streams are stable across platforms; every float input is
rounded to an fp16-representable value so fp16 and fp32 paths see identical numbers.
"""
import zlib

import numpy as np


def _rng(name):
    return np.random.default_rng(zlib.crc32(name.encode()))


def _h(x):
    """round to fp16-representable fp32"""
    return x.astype(np.float16).astype(np.float32)


def region_table(rng, Bw, L, S, nreg=3, differ=True):
    """A table shaped like encode_region_map_sp's output (encode_region_map_function.py:49-69):
    per region r, columns cols_r get +weight_r inside the mask and -outside_r elsewhere."""
    w = np.zeros((Bw, L, S), dtype=np.float32)
    weights = [0.5, 0.75, 1.0, 0.25]
    outs = [0.0, 0.25, 0.5, 0.0]
    for r in range(nreg):
        mask = rng.random(L) < 0.4
        val = np.where(mask, weights[r % 4], -outs[r % 4]).astype(np.float32)
        for c in (2 + 2 * r, 3 + 2 * r):
            if c < S:
                w[:, :, c] += val[None, :]
    if differ and Bw > 1:           # exercise the b = bh // H indexing with distinct rows
        w[1] *= 1.5
    return w


def attn_inputs(name, Bc, H, L, S, d, Bw=None, nrows=16):
    rng = _rng("attn/" + name)
    q = _h(rng.standard_normal((Bc, H, L, d)))
    k = _h(rng.standard_normal((Bc, H, S, d)))
    v = _h(rng.standard_normal((Bc, H, S, d)))
    w = region_table(rng, Bc if Bw is None else Bw, L, S)
    rows = np.sort(rng.choice(L, size=min(nrows, L), replace=False)).astype(np.int64)
    return {"q": q, "k": k, "v": v, "w": w, "rows": rows}


def mask_inputs(L=256, S=77, BH=16):
    """additive attention masks for the region path (attention_modify.py:85-91,144): fp16-representable, a few strongly
    negative entries; `ls` broadcasts into [L, S], `s1` is [1, S], `bh1s` is the [B*H, 1, S] shape prepare_attention_mask
    returns, `b4` a 4-D mask that the reference's in-place add refuses"""
    rng = _rng("masks")
    m = {"ls": _h(rng.standard_normal((L, S)) * 1.5), "s1": _h(rng.standard_normal((1, S)) * 2.0),
         "bh1s": _h(rng.standard_normal((BH, 1, S)) * 1.5)}
    m["ls"][:, 70:] = -8.0
    m["s1"][:, 5] = -20.0
    m["bh1s"][:, :, 60:] = -6.0
    m["bool"] = rng.random((L, S)) < 0.7
    return m


def proc_inputs():
    rng = _rng("proc")
    C, H, ctx, L, S = 160, 4, 96, 64, 77
    p = {"C": C, "H": H, "ctx": ctx, "L": L, "S": S}
    p["hidden"] = _h(rng.standard_normal((2, L, C)))
    p["enc"] = _h(rng.standard_normal((2, S, ctx)))
    p["wq"] = _h(rng.standard_normal((C, C)) / np.sqrt(C))
    p["wk"] = _h(rng.standard_normal((C, ctx)) / np.sqrt(ctx))
    p["wv"] = _h(rng.standard_normal((C, ctx)) / np.sqrt(ctx))
    p["wk_self"] = _h(rng.standard_normal((C, C)) / np.sqrt(C))
    p["wv_self"] = _h(rng.standard_normal((C, C)) / np.sqrt(C))
    p["wo"] = _h(rng.standard_normal((C, C)) / np.sqrt(C))
    p["bo"] = _h(rng.standard_normal((C,)) * 0.1)
    p["w"] = region_table(rng, 2, L, S)
    return p


def ip_inputs():
    """IP-Adapter operands on top of proc_inputs(): two adapters (4 and 16 image tokens, scales 0.6 / 1.1) with their
    to_k_ip / to_v_ip weights ([C, ctx], bias-free like attention_modify.py:241-246,545-550)."""
    rng = _rng("ip")
    p = proc_inputs()
    C, ctx = p["C"], p["ctx"]
    q = {"num_tokens": (4, 16), "scale": [0.6, 1.1]}
    for i, T in enumerate(q["num_tokens"]):
        q[f"ip{i}"] = _h(rng.standard_normal((2, T, ctx)))
        q[f"wk_ip{i}"] = _h(rng.standard_normal((C, ctx)) / np.sqrt(ctx))
        q[f"wv_ip{i}"] = _h(rng.standard_normal((C, ctx)) / np.sqrt(ctx))
    return q


# ----------------------------------------------------------------------------- region encoder inputs
_VOCAB = {}


class FakeTokenizer:
    """Deterministic word -> id tokenizer with the call surface encode_region_map_sp uses
    (encode_region_map_function.py:42-47)."""
    model_max_length = 77

    @staticmethod
    def word_id(word):
        return 1000 + (zlib.crc32(word.encode()) % 40000)

    def __call__(self, text, max_length=None, truncation=True, add_special_tokens=False, **kw):
        ids = [self.word_id(t) for t in text.split()]
        if add_special_tokens:
            ids = [49406] + ids + [49407]
        if max_length is not None and truncation:
            ids = ids[:max_length]
        return type("Enc", (), {"input_ids": ids})()


def prompt_ids(prompt, S=77):
    """[1,S] CLIP-style ids: BOS, words, EOS padding."""
    tok = FakeTokenizer()
    ids = [49406] + tok(prompt).input_ids
    ids = ids[: S - 1]
    ids = ids + [49407] * (S - len(ids))
    return np.array([ids], dtype=np.int64)


def rect_map(H, W, x0, y0, x1, y1):
    """uint8 map, region = pixels < 255 (encode_region_map_function.py:49); coordinates in 64-px cells."""
    m = np.full((H, W), 255, dtype=np.uint8)
    m[y0 * 64:y1 * 64, x0 * 64:x1 * 64] = 0
    return m


def region_state_inputs():
    cases = {}
    P = "a photo of a red apple on a wooden table near a blue vase and a red apple"
    N = "blurry low quality"
    ids = [prompt_ids(N), prompt_ids(P)]
    # 1 region, UI defaults weight 0.5 / mask_outsides 0 (app.py:1332-1336)
    cases["r1_512"] = ({"wooden table": {"map": rect_map(512, 512, 0, 4, 8, 8), "weight": 0.5, "mask_outsides": 0.0}},
                       ids, 512, 512, 1)
    # 2 regions; "red apple" occurs twice in the prompt; non-zero outside penalty
    cases["r2_512"] = ({"red apple": {"map": rect_map(512, 512, 1, 1, 4, 5), "weight": 0.5, "mask_outsides": 0.3},
                        "blue vase": {"map": rect_map(512, 512, 5, 0, 8, 4), "weight": 0.8, "mask_outsides": 0.0}},
                       ids, 512, 512, 1)
    # 4 regions, two images per prompt
    cases["r4_512_n2"] = ({"red apple": {"map": rect_map(512, 512, 0, 0, 4, 4), "weight": 0.5, "mask_outsides": 0.0},
                           "blue vase": {"map": rect_map(512, 512, 4, 0, 8, 4), "weight": 0.5, "mask_outsides": 0.0},
                           "wooden table": {"map": rect_map(512, 512, 0, 4, 4, 8), "weight": 0.5, "mask_outsides": 0.0},
                           "photo": {"map": rect_map(512, 512, 4, 4, 8, 8), "weight": 0.5, "mask_outsides": 0.1}},
                          ids, 512, 512, 2)
    # quirk q3: an empty region (no pixel < 255) makes `== max` all-true -> whole level positive
    cases["empty_region"] = ({"blue vase": {"map": np.full((512, 512), 255, np.uint8), "weight": 0.5, "mask_outsides": 0.2}},
                             ids, 512, 512, 1)
    # quirk q2: state None still yields a dict of zeros
    cases["state_none"] = (None, ids, 512, 512, 1)
    # text ids None -> non-dict
    cases["ids_none"] = (None, [None, None], 512, 512, 1)
    # map None is skipped; phrase not in the prompt prints and leaves zeros
    cases["map_none_notfound"] = ({"red apple": {"map": None, "weight": 0.5, "mask_outsides": 0.0},
                                   "green dragon": {"map": rect_map(512, 512, 0, 0, 8, 8), "weight": 0.5, "mask_outsides": 0.0}},
                                  ids, 512, 512, 1)
    # non-square 768x512 (L = 96*64 at level 0)
    cases["r2_768x512"] = ({"red apple": {"map": rect_map(512, 768, 0, 0, 6, 8), "weight": 0.5, "mask_outsides": 0.0},
                            "blue vase": {"map": rect_map(512, 768, 6, 2, 12, 6), "weight": 1.0, "mask_outsides": 0.5}},
                           ids, 768, 512, 1)
    # two prompts, per-prompt state list
    P2 = "a green dragon flying over a red apple"
    ids2 = [np.concatenate([prompt_ids(N), prompt_ids(N)]), np.concatenate([prompt_ids(P), prompt_ids(P2)])]
    cases["two_prompts"] = ([{"wooden table": {"map": rect_map(512, 512, 0, 4, 8, 8), "weight": 0.5, "mask_outsides": 0.0}},
                             {"green dragon": {"map": rect_map(512, 512, 2, 0, 6, 4), "weight": 0.6, "mask_outsides": 0.0}}],
                            ids2, 512, 512, 1)
    return cases


# ----------------------------------------------------------------------------- prompt encoding inputs
class FakeClipTokenizer(FakeTokenizer):
    """FakeTokenizer with the extra surface reference prompt_parser.py uses (:225-265): list input returning
    {"input_ids": [...]}, get_vocab(), bos / eos ids; commas and brackets are tokens of their own (CLIP ids 267 etc.)."""
    bos_token_id, eos_token_id = 49406, 49407
    _PUNCT = {",": 267, "(": 7, ")": 8, "[": 9, "]": 10, ".": 269, ":": 281}

    def get_vocab(self):
        v = {k + "</w>": i for k, i in self._PUNCT.items()}
        v.update({"((</w>": 11, "))</w>": 12, "[[</w>": 13, "plain</w>": 14})
        return v

    def _ids(self, text):
        out = []
        for w in text.replace(",", " , ").replace(".", " . ").replace(":", " : ").split():
            out.append(self._PUNCT[w] if w in self._PUNCT else self.word_id(w))
        return out

    def __call__(self, text, max_length=None, truncation=True, add_special_tokens=False, **kw):
        if isinstance(text, (list, tuple)):
            return {"input_ids": [self._ids(t) for t in text]}
        ids = self._ids(text)
        if add_special_tokens:
            ids = [self.bos_token_id] + ids + [self.eos_token_id]
        if max_length is not None and truncation:
            ids = ids[:max_length]
        return type("Enc", (), {"input_ids": ids})()


def fake_text_encoder(dim=32, layers=3, seed=11):
    """Deterministic stand-in for CLIPTextModel with the attributes prompt_parser.py touches (:267-279): callable on a
    [B, 77] id tensor with output_hidden_states=True -> .last_hidden_state, .hidden_states; .text_model.final_layer_norm;
    .device / .dtype.  Token + position embedding followed by `layers` fixed tanh mixing layers."""
    import torch

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.pos = torch.randn(77, dim, generator=g) * 0.1
            self.mix = [torch.randn(dim, dim, generator=g) / dim ** 0.5 for _ in range(layers)]
            self.freq = torch.rand(dim, generator=g) * 3.0 + 0.5
            self.text_model = type("TM", (), {})()
            self.text_model.final_layer_norm = torch.nn.LayerNorm(dim)
            with torch.no_grad():      # a real CLIP has a learned affine: keeps the tensor mean away from 0 (mean restoration divides by it)
                self.text_model.final_layer_norm.bias.copy_(torch.rand(dim, generator=g) * 0.5 + 0.25)
                self.text_model.final_layer_norm.weight.copy_(torch.rand(dim, generator=g) * 0.5 + 0.75)
            self.device, self.dtype = torch.device("cpu"), torch.float32

        def forward(self, tokens, output_hidden_states=False):
            t = tokens.to(torch.float32)
            h = torch.sin(t[..., None] * 1e-3 * self.freq) + self.pos[: tokens.shape[1]]
            hs = [h]
            for m in self.mix:
                h = torch.tanh(h @ m) + 0.5 * h
                hs.append(h)
            return type("Out", (), {"last_hidden_state": self.text_model.final_layer_norm(h), "hidden_states": tuple(hs)})()

    return Enc()


def prompt_cases():
    """prompts for parse_prompt_attention / chunking / weighted encoding"""
    long_a = ", ".join(f"item{i} with detail" for i in range(30))                 # > 75 tokens with commas near the boundary
    long_b = " ".join(f"word{i}" for i in range(160))                            # > 150 tokens, no commas: hard splits
    return {
        "parse": ["normal text", "an (important) word", "(unbalanced", "\\(literal\\]", "(unnecessary)(parens)",
                  "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).", "", "a [b (c:1.5) d] e", "x:1.2) y", "tail \\",
                  "one BREAK two", "(a BREAK b:1.3) c", "]stray) text[", "(((deep))) [[[less]]] (mix [in:0.8] out)",
                  "colon: inside (a:b) and (num:+.5) (neg:-1)", "multi\nline BREAK  BREAK end"],
        "chunk": ["a red apple, on a table", long_a, long_b, "first part BREAK second part, (weighted:1.4) end",
                  "(" + long_a + ":1.2)", ""],
        "encode": [["blurry, low quality", "a (red:1.3) apple on a [wooden] table"], ["", long_a],
                   ["x BREAK y", "(a:0.5) b, c"]],
    }


# ----------------------------------------------------------------------------- samplers (tests/golden/samplers_extra.npz)
def analytic_denoiser(x, sigma, **extra_args):
    """A smooth stand-in for the CFG-combined denoiser: the exact posterior mean for N(0, 1.3^2) data plus a bounded
    non-linear term, so that every slope / stage of a sampler contributes to the result."""
    import torch
    s = sigma.reshape(-1, *([1] * (x.ndim - 1))).to(x.dtype)
    return x * (1.69 / (1.69 + s * s)) + 0.1 * torch.tanh(x) * s / (1 + s)


def sampler_cases():
    """name -> (function name in samplers_extra_k_diffusion, number of sigmas, kwargs, torch seed)"""
    return {
        "restart_25": ("restart_sampler", 25, {}, 3),                       # automatic list: one restart of 9 steps
        "restart_40": ("restart_sampler", 40, {}, 4),                       # automatic list: steps // 4, twice
        "restart_10": ("restart_sampler", 10, {}, 5),                       # no restart below 20 steps
        "restart_explicit": ("restart_sampler", 12, {"restart_list": {0.3: [4, 2, 3.0]}, "s_noise": 0.7}, 6),
        "ddpm_15": ("sample_ddpm", 15, {}, 7),
        "lcm_6": ("sample_lcm", 6, {}, 8),
        "heunpp2_12": ("sample_heunpp2", 12, {}, 9),
        "heunpp2_churn": ("sample_heunpp2", 9, {"s_churn": 4.0, "s_tmin": 0.05, "s_tmax": 10.0, "s_noise": 1.003}, 10),
    }


def sampler_start(n_sigmas):
    """(x0 [2,4,8,8] float64 scaled by sigma_max, Karras sigmas + trailing zero) for the SD1.5 sigma range"""
    import torch
    g = torch.Generator().manual_seed(1234 + n_sigmas)
    ramp = torch.linspace(0, 1, n_sigmas, dtype=torch.float64)
    lo, hi = 0.029167533 ** (1 / 7.0), 14.614646912 ** (1 / 7.0)
    sigmas = torch.cat([(hi + ramp * (lo - hi)) ** 7.0, torch.zeros(1, dtype=torch.float64)])
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64) * sigmas[0]
    return x, sigmas


# ----------------------------------------------------------------------------- long_encode 1 / 2 (tests/golden/prompt_encoders.npz)
class FakeHFClipTokenizer(FakeClipTokenizer):
    """FakeClipTokenizer with the transformers call surface reference encoder_prompt_modify.py uses for `long_encode` 1 / 2
    (:127-160, :540-560, :640-650): special tokens added, `padding` ("max_length" / "longest"), `truncation`, `return_tensors`,
    list input, `batch_decode`, `pad_token_id`."""
    pad_token_id = 49407

    def _one(self, text, max_length, truncation):
        ids = [self.bos_token_id] + self._ids(text) + [self.eos_token_id]
        if truncation and max_length is not None and len(ids) > max_length:
            ids = ids[:max_length - 1] + [self.eos_token_id]
        return ids

    def __call__(self, text, padding=False, max_length=None, truncation=False, return_tensors=None, **kw):
        import torch
        single = not isinstance(text, (list, tuple))
        rows = [self._one(t, max_length, truncation) for t in ([text] if single else text)]
        if padding == "max_length":
            rows = [r + [self.pad_token_id] * (max_length - len(r)) for r in rows]
        elif padding == "longest":
            n = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (n - len(r)) for r in rows]
        mask = [[1] * len(r) for r in rows]
        if return_tensors == "pt":
            return type("Enc", (), {"input_ids": torch.tensor(rows, dtype=torch.long), "attention_mask": torch.tensor(mask)})()
        return type("Enc", (), {"input_ids": rows[0] if single else rows, "attention_mask": mask[0] if single else mask})()

    def batch_decode(self, ids):
        return [" ".join(str(int(v)) for v in row) for row in ids]


def fake_hf_text_encoder(dim=32, layers=3, seed=11):
    """fake_text_encoder with the transformers return convention those branches index: out[0] = last hidden state (final
    LayerNorm applied), out[-1] = the tuple of hidden states when output_hidden_states=True; `attention_mask` accepted."""
    import torch
    base = fake_text_encoder(dim, layers, seed)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.text_model = base.text_model
            self.config = type("Cfg", (), {})()
            self.device, self.dtype = torch.device("cpu"), torch.float32

        def forward(self, tokens, attention_mask=None, output_hidden_states=False):
            o = base(tokens, output_hidden_states=True)
            pooled = o.last_hidden_state[:, -1]
            return (o.last_hidden_state, pooled, o.hidden_states) if output_hidden_states else (o.last_hidden_state, pooled)

    return Enc()


def prompt_encoder_cases():
    """(negative, prompt) pairs for the lpw-style (`long_encode=1`) and plain (`long_encode=2`) encoders"""
    long_a = ", ".join(f"item{i} with detail" for i in range(30))                 # ~150 tokens: three chunks after weighting
    return [("blurry, low quality", "a (red:1.3) apple on a [wooden] table"), ("", long_a),
            ("(worst quality:1.4), lowres", "((masterpiece)), (" + long_a + ":0.9), [background]"),
            ("text, watermark", "plain prompt without weights")]
