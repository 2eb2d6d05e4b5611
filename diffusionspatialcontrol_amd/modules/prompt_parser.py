"""Prompt emphasis parsing and unlimited-length (75-token chunk) weighted CLIP encoding - the build's counterpart of
reference `source/modules/prompt_parser.py` (`parse_prompt_attention` :303-383, `FrozenCLIPEmbedderWithCustomWords`
:22-279), the default prompt path of the pipelines (`encoder_prompt_modify.py:691-812`, `long_encode == 0`).

Host-side text processing: no device kernels here.  The behaviour - including its quirks - is
pinned by goldens captured from the reference's own module on a deterministic fake tokenizer / text encoder
(tests/golden/prompt_parser.npz): emphasis syntax `(a)`, `(a:1.3)`, `[a]`, escapes, unbalanced brackets, `BREAK`,
comma back-tracking at chunk borders, per-token multipliers with mean restoration.

Written from the behaviour, not from the reference's text: a single left-to-right scanner replaces its regular expression.
"""
import math
import re

import numpy as np
import torch

_UP, _DOWN = 1.1, 1 / 1.1
_BREAK_SPLIT = re.compile(r"\s*\bBREAK\b\s*", re.S)
_ESCAPABLE = "()[]\\"
_NUMBER_CHARS = ".0123456789"


def _weight_suffix(text, i):
    """`:<number>)` starting at text[i] == ':' -> (number string, index after ')') or None.
    <number> = optional sign, then one or more characters of '.' and digits (validated by float() later)."""
    j = i + 1
    if j < len(text) and text[j] in "+-":
        j += 1
    k = j
    while k < len(text) and text[k] in _NUMBER_CHARS:
        k += 1
    if k == j or k >= len(text) or text[k] != ")":
        return None
    return text[i + 1:k], k + 1


def parse_prompt_attention(text):
    """'a (b:1.3) [c]' -> [['a ', 1.0], ['b', 1.3], [' ', 1.0], ['c', 0.909...]]; `BREAK` words become ['BREAK', -1]
    entries.  See the reference docstring (:304-337) for the accepted syntax; results are identical to it."""
    pieces = []                       # [text, weight]
    open_round, open_square = [], []  # positions in `pieces` where an open bracket started

    def scale_from(start, factor):
        for p in pieces[start:]:
            p[1] *= factor

    def plain(s):
        for n, part in enumerate(_BREAK_SPLIT.split(s)):
            if n:
                pieces.append(["BREAK", -1])
            pieces.append([part, 1.0])

    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch == "\\":
            if i + 1 < n and text[i + 1] in _ESCAPABLE:
                pieces.append([text[i + 1], 1.0])
                i += 2
            else:                                     # a lone backslash vanishes (leaves an empty piece)
                pieces.append(["", 1.0])
                i += 1
        elif ch == "(":
            open_round.append(len(pieces))
            i += 1
        elif ch == "[":
            open_square.append(len(pieces))
            i += 1
        elif ch == ":":
            hit = _weight_suffix(text, i)
            if hit is None:
                plain(":")
                i += 1
            elif open_round:
                scale_from(open_round.pop(), float(hit[0]))
                i = hit[1]
            else:                                     # a weight with nothing to close: ordinary text
                plain(text[i:hit[1]])
                i = hit[1]
        elif ch == ")":
            if open_round:
                scale_from(open_round.pop(), _UP)
            else:
                plain(")")
            i += 1
        elif ch == "]":
            if open_square:
                scale_from(open_square.pop(), _DOWN)
            else:
                plain("]")
            i += 1
        else:
            j = i
            while j < n and text[j] not in "\\()[]:":
                j += 1
            plain(text[i:j])
            i = j
    for start in open_round:                          # unbalanced brackets still count
        scale_from(start, _UP)
    for start in open_square:
        scale_from(start, _DOWN)
    if not pieces:
        pieces = [["", 1.0]]
    merged = [pieces[0]]
    for t, w in pieces[1:]:
        if w == merged[-1][1]:
            merged[-1][0] += t
        else:
            merged.append([t, w])
    return merged


class PromptChunk:
    """77 token ids (BOS + 75 + EOS) with one multiplier per token"""

    def __init__(self, tokens=None, multipliers=None):
        self.tokens = tokens if tokens is not None else []
        self.multipliers = multipliers if multipliers is not None else []
        self.fixes = []


class FrozenCLIPEmbedderWithCustomWordsBase(torch.nn.Module):
    """Chunking + weighting around a text encoder (reference :22-221).  Subclasses provide `tokenize`,
    `encode_with_transformers`, `id_start`, `id_end`, `id_pad`, `comma_token`."""

    chunk_length = 75
    comma_padding_backtrack = 20      # webui default (reference :96)

    def __init__(self, text_encoder, enable_emphasis=True):
        super().__init__()
        self.device = lambda: text_encoder.device
        self.enable_emphasis = enable_emphasis

    def empty_chunk(self):
        return PromptChunk([self.id_start] + [self.id_end] * (self.chunk_length + 1), [1.0] * (self.chunk_length + 2))

    def get_target_prompt_token_count(self, token_count):
        return math.ceil(max(token_count, 1) / self.chunk_length) * self.chunk_length

    def _seal(self, tokens, mults):
        pad = self.chunk_length - len(tokens)
        return PromptChunk([self.id_start] + tokens + [self.id_end] * pad + [self.id_end], [1.0] + mults + [1.0] * pad + [1.0])

    def tokenize_line(self, line):
        """one prompt -> (list of PromptChunk, token count).  A chunk that fills up is closed; if a comma lies within the
        last `comma_padding_backtrack` tokens, everything after that comma moves to the next chunk instead of being cut."""
        parsed = parse_prompt_attention(line) if self.enable_emphasis else [[line, 1.0]]
        tokenized = self.tokenize([t for t, _ in parsed])
        L = self.chunk_length
        chunks, count = [], 0
        cur_t, cur_m, last_comma = [], [], -1

        def close(final=False):
            nonlocal cur_t, cur_m, last_comma, count
            count += len(cur_t) if final else L
            chunks.append(self._seal(cur_t, cur_m))
            cur_t, cur_m, last_comma = [], [], -1

        for ids, (text, weight) in zip(tokenized, parsed):
            if text == "BREAK" and weight == -1:
                close()
                continue
            for tok in ids:
                if tok == self.comma_token:
                    last_comma = len(cur_t)
                elif (self.comma_padding_backtrack and len(cur_t) == L and last_comma != -1
                      and len(cur_t) - last_comma <= self.comma_padding_backtrack):
                    cut = last_comma + 1
                    moved_t, moved_m = cur_t[cut:], cur_m[cut:]
                    cur_t, cur_m = cur_t[:cut], cur_m[:cut]
                    close()
                    cur_t, cur_m = moved_t, moved_m
                if len(cur_t) == L:
                    close()
                cur_t.append(tok)
                cur_m.append(weight)
        if cur_t or not chunks:
            close(final=True)
        return chunks, count

    def process_texts(self, texts):
        seen, batch, most = {}, [], 0
        for line in texts:
            if line not in seen:
                seen[line], n = self.tokenize_line(line)
                most = max(most, n)
            batch.append(seen[line])
        return batch, most

    def forward(self, texts):
        """texts -> (ids [B, 77 k] numpy, embeddings [B, 77 k, C]); every text is padded to the longest one's chunk count"""
        batch, _ = self.process_texts(texts)
        n_chunks = max(len(c) for c in batch)
        ids, zs = [], []
        for k in range(n_chunks):
            row = [c[k] if k < len(c) else self.empty_chunk() for c in batch]
            tokens = [c.tokens for c in row]
            zs.append(self.process_tokens(tokens, [c.multipliers for c in row]))
            ids.append(tokens)
        return np.hstack(ids), torch.hstack(zs)

    def process_tokens(self, remade_batch_tokens, batch_multipliers):
        """one 77-token chunk per text through the encoder, then z * multiplier with the tensor mean restored (:196-221)"""
        tokens = torch.asarray(remade_batch_tokens).to(self.device())
        if self.id_end != self.id_pad:                    # SD2-style tokenizers pad with a different id than EOS
            for b, row in enumerate(remade_batch_tokens):
                tokens[b, row.index(self.id_end) + 1:] = self.id_pad
        z = self.encode_with_transformers(tokens)
        m = torch.asarray(batch_multipliers).to(self.device())
        before = z.mean()
        z = z * m.reshape(m.shape + (1,)).expand(z.shape)
        return z * (before / z.mean())


class FrozenCLIPEmbedderWithCustomWords(FrozenCLIPEmbedderWithCustomWordsBase):
    """transformers CLIP tokenizer + `CLIPTextModel` (reference :224-279); `CLIP_stop_at_last_layers` > 1 takes an earlier
    hidden state through the final LayerNorm (clip skip)."""

    def __init__(self, tokenizer, text_encoder, CLIP_stop_at_last_layers):
        super().__init__(text_encoder)
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.CLIP_stop_at_last_layers = CLIP_stop_at_last_layers
        vocab = tokenizer.get_vocab()
        self.comma_token = vocab.get(",</w>", None)
        self.token_mults = {}
        for word, ident in vocab.items():                 # kept for interface parity: tokens that contain brackets
            if any(c in word for c in "()[]"):
                mult = 1.0
                for c in word:
                    if c in "(]":
                        mult *= 1.1
                    elif c in "[)":
                        mult /= 1.1
                if mult != 1.0:
                    self.token_mults[ident] = mult
        self.id_start, self.id_end = tokenizer.bos_token_id, tokenizer.eos_token_id
        self.id_pad = self.id_end

    def tokenize(self, texts):
        return self.tokenizer(texts, truncation=False, add_special_tokens=False)["input_ids"]

    def encode_with_transformers(self, tokens):
        out = self.text_encoder(tokens.to(self.text_encoder.device), output_hidden_states=True)
        skip = self.CLIP_stop_at_last_layers
        if skip is not None and skip > 1:
            return self.text_encoder.text_model.final_layer_norm(out.hidden_states[-skip])
        return out.last_hidden_state
