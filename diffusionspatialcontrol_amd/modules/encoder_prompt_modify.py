"""Prompt -> (embeddings, token ids) for the pipelines - counterpart of reference `source/modules/encoder_prompt_modify.py`
`encode_prompt_function` (:814-832) and its default branch `encode_prompt_automatic1111` (:691-812): one
`FrozenCLIPEmbedderWithCustomWords([negative, prompt])` call per prompt (both texts padded to the same number of
77-token chunks), embeddings repeated per image, the token ids returned for the region encoder.
The two other branches are here too: `long_encode == 1` -> `encoder_long_prompt` (:395-490: "long prompt weighting" - parse
emphasis, tokenise word by word, pad to a multiple of 75 tokens (at most `max_embeddings_multiples` chunks), encode chunk by
chunk, multiply by the token weights, restore the tensor mean; :41-393), anything else -> `encode_short_prompt` (:492-689: plain
77-token CLIP with truncation).  Both pinned by goldens captured from the reference's own functions on a deterministic fake
tokenizer / encoder (tests/golden/make_golden_prompts.py).  LoRA scaling of the text encoder and textual-inversion token
expansion are not built."""
import re

import numpy as np
import torch

from .prompt_parser import FrozenCLIPEmbedderWithCustomWords


def encode_prompt_automatic1111(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                                prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
    if lora_scale is not None:
        raise NotImplementedError("LoRA scaling of the text encoder is outside the path built here")
    if prompt is not None and isinstance(prompt, str):
        batch_size = 1
    elif prompt is not None and isinstance(prompt, list):
        batch_size = len(prompt)
    else:
        batch_size = prompt_embeds.shape[0]
    uncond = []
    if do_classifier_free_guidance and negative_prompt_embeds is None:
        if negative_prompt is None:
            uncond = [""] * batch_size
        elif prompt is not None and type(prompt) is not type(negative_prompt):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                            f" {type(prompt)}.")
        elif isinstance(negative_prompt, str):
            uncond = [negative_prompt] + [""] * (batch_size - 1)
        elif batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                             f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                             " the batch size of `prompt`.")
        else:
            uncond = negative_prompt
    if len(uncond) == 0:
        uncond = [""] * batch_size
    if prompt_embeds is None and not isinstance(prompt, list):
        prompt = [prompt]
    parser = FrozenCLIPEmbedderWithCustomWords(pipe.tokenizer, pipe.text_encoder, clip_skip)
    pos_e, neg_e, pos_ids, neg_ids = [], [], [], []
    for i in range(batch_size):
        ids, emb = parser([uncond[i], prompt[i]])                   # both padded to the same chunk count (:766)
        n_emb, p_emb = torch.chunk(emb, 2, dim=0)
        n_id, p_id = np.split(ids, ids.shape[0])
        pos_e.append(p_emb); neg_e.append(n_emb); pos_ids.append(p_id); neg_ids.append(n_id)
    prompt_tokens_id = negative_prompt_tokens_id = None
    if prompt_embeds is None:
        prompt_embeds = torch.cat(pos_e)
        prompt_tokens_id = np.concatenate(pos_ids)
    if do_classifier_free_guidance and negative_prompt_embeds is None:
        negative_prompt_embeds = torch.cat(neg_e)
        negative_prompt_tokens_id = np.concatenate(neg_ids)
    if pipe.text_encoder is not None:
        dtype = pipe.text_encoder.dtype
    elif pipe.unet is not None:
        dtype = pipe.unet.dtype
    else:
        dtype = prompt_embeds.dtype
    prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
    b, s, _ = prompt_embeds.shape
    prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, s, -1)
    if do_classifier_free_guidance:
        s = negative_prompt_embeds.shape[1]
        negative_prompt_embeds = negative_prompt_embeds.to(dtype=dtype, device=device)
        negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
            batch_size * num_images_per_prompt, s, -1)
    return prompt_embeds, negative_prompt_embeds, [negative_prompt_tokens_id, prompt_tokens_id]


# ----------------------------------------------------------------------------------------------- long_encode == 1
_LPW_TOKEN = re.compile(r"\\\(|\\\)|\\\[|\\]|\\\\|\\|\(|\[|:([+-]?[.\d]+)\)|\)|]|[^\\()\[\]:]+|:")


def parse_prompt_attention(text):
    """reference :41-125 (the lpw variant: no BREAK keyword): [[text, weight], ...] with `(a)` x1.1, `(a:w)` xw, `[a]` /1.1,
    backslash escapes, unbalanced brackets closed at the end, neighbours of equal weight merged"""
    out, round_open, square_open = [], [], []

    def scale_from(pos, m):
        for item in out[pos:]:
            item[1] *= m

    for m in _LPW_TOKEN.finditer(text):
        tok, w = m.group(0), m.group(1)
        if tok.startswith("\\"):
            out.append([tok[1:], 1.0])
        elif tok == "(":
            round_open.append(len(out))
        elif tok == "[":
            square_open.append(len(out))
        elif w is not None and round_open:
            scale_from(round_open.pop(), float(w))
        elif tok == ")" and round_open:
            scale_from(round_open.pop(), 1.1)
        elif tok == "]" and square_open:
            scale_from(square_open.pop(), 1 / 1.1)
        else:
            out.append([tok, 1.0])
    for pos in round_open:
        scale_from(pos, 1.1)
    for pos in square_open:
        scale_from(pos, 1 / 1.1)
    if not out:
        out = [["", 1.0]]
    merged = [out[0]]
    for item in out[1:]:
        if item[1] == merged[-1][1]:
            merged[-1][0] += item[0]
        else:
            merged.append(item)
    return merged


def get_prompts_with_weights(pipe, prompt, max_length):
    """reference :127-160: per prompt the token ids (no BOS / EOS) and one weight per token, cut at max_length"""
    tokens, weights = [], []
    for text in prompt:
        ids, ws = [], []
        for word, weight in parse_prompt_attention(text):
            tk = pipe.tokenizer(word).input_ids[1:-1]
            ids += tk
            ws += [weight] * len(tk)
            if len(ids) > max_length:
                break
        tokens.append(ids[:max_length])
        weights.append(ws[:max_length])
    return tokens, weights


def pad_tokens_and_weights(tokens, weights, max_length, bos, eos, pad, no_boseos_middle=True, chunk_length=77):
    """reference :162-184: BOS + tokens + padding + EOS; the weights get 1.0 for BOS / EOS (of every chunk unless
    no_boseos_middle) and for the padding"""
    n_chunks = (max_length - 2) // (chunk_length - 2)
    per = chunk_length - 2
    for i in range(len(tokens)):
        tokens[i] = [bos] + tokens[i] + [pad] * (max_length - 2 - len(tokens[i])) + [eos]
        if no_boseos_middle:
            weights[i] = [1.0] + weights[i] + [1.0] * (max_length - 1 - len(weights[i]))
        elif len(weights[i]) == 0:
            weights[i] = [1.0] * (n_chunks * chunk_length)
        else:
            w = []
            for j in range(n_chunks):
                w += [1.0] + weights[i][j * per:min(len(weights[i]), (j + 1) * per)] + [1.0]
            weights[i] = w + [1.0] * (n_chunks * chunk_length - len(w))
    return tokens, weights


def clip_skip_prompt(pipe, text_input, clip_skip=None):
    """reference :186-210: last hidden state, or the hidden state `clip_skip` layers from the end + the final LayerNorm"""
    if clip_skip is not None and clip_skip > 1:
        hidden = pipe.text_encoder(text_input, attention_mask=None, output_hidden_states=True)[-1][-clip_skip]
        return pipe.text_encoder.text_model.final_layer_norm(hidden)
    return pipe.text_encoder(text_input, attention_mask=None)[0]


def get_unweighted_text_embeddings(pipe, text_input, chunk_length, no_boseos_middle=True, clip_skip=None):
    """reference :212-252: encode 77-token windows (75 payload tokens each, BOS / EOS of the whole input re-attached)"""
    n_chunks = (text_input.shape[1] - 2) // (chunk_length - 2)
    if n_chunks <= 1:
        return clip_skip_prompt(pipe, text_input, clip_skip)
    per = chunk_length - 2
    parts = []
    for i in range(n_chunks):
        chunk = text_input[:, i * per:(i + 1) * per + 2].clone()
        chunk[:, 0] = text_input[0, 0]
        chunk[:, -1] = text_input[0, -1]
        emb = clip_skip_prompt(pipe, chunk, clip_skip)
        if no_boseos_middle:
            emb = emb[:, :-1] if i == 0 else (emb[:, 1:] if i == n_chunks - 1 else emb[:, 1:-1])
        parts.append(emb)
    return torch.cat(parts, dim=1)


def get_weighted_text_embeddings(pipe, prompt, uncond_prompt=None, max_embeddings_multiples=3, no_boseos_middle=False,
                                 skip_parsing=False, skip_weighting=False, clip_skip=None):
    """reference :254-393 -> (text embeddings, uncond embeddings | None, uncond ids | None, prompt ids)"""
    mml = pipe.tokenizer.model_max_length
    max_length = (mml - 2) * max_embeddings_multiples + 2
    prompt = [prompt] if isinstance(prompt, str) else prompt
    if uncond_prompt is not None and isinstance(uncond_prompt, str):
        uncond_prompt = [uncond_prompt]

    def tokens_of(texts):
        if not skip_parsing:
            return get_prompts_with_weights(pipe, texts, max_length - 2)
        ids = [t[1:-1] for t in pipe.tokenizer(texts, max_length=max_length, truncation=True).input_ids]
        return ids, [[1.0] * len(t) for t in ids]

    p_tok, p_w = tokens_of(prompt)
    u_tok = u_w = None
    if uncond_prompt is not None:
        u_tok, u_w = tokens_of(uncond_prompt)
    longest = max(len(t) for t in p_tok + (u_tok or []))
    max_embeddings_multiples = max(1, min(max_embeddings_multiples, (longest - 1) // (mml - 2) + 1))
    max_length = (mml - 2) * max_embeddings_multiples + 2
    bos, eos = pipe.tokenizer.bos_token_id, pipe.tokenizer.eos_token_id
    pad = getattr(pipe.tokenizer, "pad_token_id", eos)
    dev = pipe.device

    def embed(tok, w):
        tok, w = pad_tokens_and_weights(tok, w, max_length, bos, eos, pad, no_boseos_middle=no_boseos_middle, chunk_length=mml)
        ids = np.array(tok, dtype=np.int64)
        emb = get_unweighted_text_embeddings(pipe, torch.tensor(tok, dtype=torch.long, device=dev), mml,
                                             no_boseos_middle=no_boseos_middle, clip_skip=clip_skip)
        if not skip_parsing and not skip_weighting:          # weight, then restore the mean of the whole tensor
            wt = torch.tensor(w, dtype=emb.dtype, device=emb.device)
            before = emb.float().mean(dim=[-2, -1]).to(emb.dtype)
            emb = emb * wt.unsqueeze(-1)
            after = emb.float().mean(dim=[-2, -1]).to(emb.dtype)
            emb = emb * (before / after).unsqueeze(-1).unsqueeze(-1)
        return emb, ids

    text_emb, prompt_ids = embed(p_tok, p_w)
    if uncond_prompt is None:
        return text_emb, None, None, prompt_ids
    uncond_emb, uncond_ids = embed(u_tok, u_w)
    return text_emb, uncond_emb, uncond_ids, prompt_ids


def _batch_size(prompt, prompt_embeds):
    if prompt is not None and isinstance(prompt, str):
        return 1
    if prompt is not None and isinstance(prompt, list):
        return len(prompt)
    return prompt_embeds.shape[0]


def encoder_long_prompt(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                        prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None,
                        max_embeddings_multiples=3):
    """reference :395-490"""
    if lora_scale is not None:
        raise NotImplementedError("LoRA scaling of the text encoder is outside the path built here")
    batch_size = _batch_size(prompt, prompt_embeds)
    neg_ids = pos_ids = None
    if negative_prompt_embeds is None:
        if negative_prompt is None:
            negative_prompt = [""] * batch_size
        elif isinstance(negative_prompt, str):
            negative_prompt = [negative_prompt] * batch_size
        if batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                             f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                             " the batch size of `prompt`.")
    if prompt_embeds is None or negative_prompt_embeds is None:
        pe, ne, neg_ids, pos_ids = get_weighted_text_embeddings(
            pipe, prompt, negative_prompt if do_classifier_free_guidance else None,
            max_embeddings_multiples=int(max_embeddings_multiples), clip_skip=clip_skip)
        prompt_embeds = pe if prompt_embeds is None else prompt_embeds
        negative_prompt_embeds = ne if negative_prompt_embeds is None else negative_prompt_embeds
    b, sl, _ = prompt_embeds.shape
    prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, sl, -1)
    if do_classifier_free_guidance:
        b, sl, _ = negative_prompt_embeds.shape
        negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, sl, -1)
    return prompt_embeds, negative_prompt_embeds, [neg_ids, pos_ids]


# ----------------------------------------------------------------------------------------------- long_encode == 2 (any other value)
def encode_short_prompt(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                        prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
    """reference :492-689: one 77-token CLIP call per text, longer prompts are truncated"""
    if lora_scale is not None:
        raise NotImplementedError("LoRA scaling of the text encoder is outside the path built here")
    batch_size = _batch_size(prompt, prompt_embeds)
    tok = pipe.tokenizer
    pos_ids = neg_ids = None

    def encode(ids):
        if clip_skip is not None and clip_skip > 1:
            hidden = pipe.text_encoder(ids.to(device), attention_mask=None, output_hidden_states=True)[-1][-clip_skip]
            return pipe.text_encoder.text_model.final_layer_norm(hidden)
        return pipe.text_encoder(ids.to(device), attention_mask=None)[0]

    if prompt_embeds is None:
        inp = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        pos_ids = inp.input_ids.detach().cpu().numpy()
        prompt_embeds = encode(inp.input_ids)
    if pipe.text_encoder is not None:
        dtype = pipe.text_encoder.dtype
    elif pipe.unet is not None:
        dtype = pipe.unet.dtype
    else:
        dtype = prompt_embeds.dtype
    prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
    b, sl, _ = prompt_embeds.shape
    prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, sl, -1)
    if do_classifier_free_guidance and negative_prompt_embeds is None:
        if negative_prompt is None:
            uncond = [""] * batch_size
        elif prompt is not None and type(prompt) is not type(negative_prompt):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                            f" {type(prompt)}.")
        elif isinstance(negative_prompt, str):
            uncond = [negative_prompt]
        elif batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                             f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                             " the batch size of `prompt`.")
        else:
            uncond = negative_prompt
        inp = tok(uncond, padding="max_length", max_length=prompt_embeds.shape[1], truncation=True, return_tensors="pt")
        neg_ids = inp.input_ids.detach().cpu().numpy()
        negative_prompt_embeds = encode(inp.input_ids)
    if do_classifier_free_guidance:
        sl = negative_prompt_embeds.shape[1]
        negative_prompt_embeds = negative_prompt_embeds.to(dtype=dtype, device=device)
        negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
            batch_size * num_images_per_prompt, sl, -1)
    return prompt_embeds, negative_prompt_embeds, [neg_ids, pos_ids]


def encode_prompt_function(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                           prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None, long_encode=False):
    if long_encode == 0:
        return encode_prompt_automatic1111(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance,
                                           negative_prompt, prompt_embeds, negative_prompt_embeds, lora_scale, clip_skip)
    if long_encode == 1:
        return encoder_long_prompt(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt,
                                   prompt_embeds, negative_prompt_embeds, lora_scale, clip_skip)
    return encode_short_prompt(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt,
                               prompt_embeds, negative_prompt_embeds, lora_scale, clip_skip)
