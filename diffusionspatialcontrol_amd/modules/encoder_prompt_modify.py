"""Prompt -> (embeddings, token ids) for the pipelines - counterpart of reference `source/modules/encoder_prompt_modify.py`
`encode_prompt_function` (:814-832) and its default branch `encode_prompt_automatic1111` (:691-812): one
`FrozenCLIPEmbedderWithCustomWords([negative, prompt])` call per prompt (both texts padded to the same number of
77-token chunks), embeddings repeated per image, the token ids returned for the region encoder.
The two other branches (`long_encode` 1 = lpw-style weighting :395-490, 2 = plain 77-token CLIP :492-689), LoRA scaling
and textual-inversion token expansion are not built."""
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


def encode_prompt_function(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                           prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None, long_encode=False):
    if long_encode == 0:
        return encode_prompt_automatic1111(pipe, prompt, device, num_images_per_prompt, do_classifier_free_guidance,
                                           negative_prompt, prompt_embeds, negative_prompt_embeds, lora_scale, clip_skip)
    raise NotImplementedError("long_encode 1 (lpw weighting) and 2 (plain 77-token CLIP) are not built; the default "
                              "A1111-style encoder (long_encode=0) is")
