#!/usr/bin/env python3
"""Capture golden vectors for the `long_encode` 1 / 2 prompt encoders by running the REFERENCE's own
source/modules/encoder_prompt_modify.py (`encoder_long_prompt` :395-490 with `get_weighted_text_embeddings` :254-393,
`encode_short_prompt` :492-689) in the build container:

    python tests/golden/make_golden_prompts.py         # rewrites tests/golden/prompt_encoders.npz

Inert stand-ins are registered for the diffusers names the file imports (type checks / LoRA scaling hooks, none of them on
the captured arithmetic); `modules.prompt_parser` is the reference's own file.  Tokenizer and text encoder are the
deterministic fakes of tests/golden/inputs.py (no CLIP vocabulary or weights exist offline).  Only data is written."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from inputs import FakeHFClipTokenizer, fake_hf_text_encoder, prompt_encoder_cases  # noqa: E402

REF = "/root/reference/source/modules"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load(fname, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, fname))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def main():
    class _Logger:
        def warning(self, *a, **k):
            pass
        info = debug = error = warning
    noop = lambda *a, **k: None  # noqa: E731
    _mod("diffusers", DiffusionPipeline=type("DiffusionPipeline", (), {}))
    _mod("diffusers.models")
    _mod("diffusers.models.lora", adjust_lora_scale_text_encoder=noop)
    _mod("diffusers.loaders", FromSingleFileMixin=type("A", (), {}), LoraLoaderMixin=type("B", (), {}),
         TextualInversionLoaderMixin=type("C", (), {}))
    _mod("diffusers.utils", USE_PEFT_BACKEND=True, deprecate=noop, logging=type("L", (), {"get_logger": staticmethod(lambda n: _Logger())})(),
         replace_example_docstring=lambda *_a, **_k: (lambda f: f), scale_lora_layers=noop, unscale_lora_layers=noop)
    _mod("modules")
    load("prompt_parser.py", "modules.prompt_parser")
    ep = load("encoder_prompt_modify.py", "ref_encoder_prompt_modify")
    if not hasattr(ep, "logger"):
        ep.logger = _Logger()
    pipe = types.SimpleNamespace(tokenizer=FakeHFClipTokenizer(), text_encoder=fake_hf_text_encoder(), device=torch.device("cpu"),
                                 unet=None)
    out = {}
    with torch.no_grad():
        for i, (neg, pos) in enumerate(prompt_encoder_cases()):
            for clip_skip in (None, 2):
                tag = f"{i}/skip{clip_skip or 0}"
                pe, ne, ids = ep.encoder_long_prompt(pipe, pos, "cpu", 2, True, neg, clip_skip=clip_skip)
                out[f"long/{tag}/pos"], out[f"long/{tag}/neg"] = pe.numpy(), ne.numpy()
                out[f"long/{tag}/neg_ids"], out[f"long/{tag}/pos_ids"] = ids[0], ids[1]
                pe, ne, ids = ep.encode_short_prompt(pipe, pos, "cpu", 2, True, neg, clip_skip=clip_skip)
                out[f"short/{tag}/pos"], out[f"short/{tag}/neg"] = pe.numpy(), ne.numpy()
                out[f"short/{tag}/neg_ids"], out[f"short/{tag}/pos_ids"] = ids[0], ids[1]
        pe, ne, ids = ep.encoder_long_prompt(pipe, prompt_encoder_cases()[0][1], "cpu", 1, False)
        out["long/nocfg/pos"], out["long/nocfg/pos_ids"] = pe.numpy(), ids[1]
        assert ne is None and ids[0] is None
    np.savez_compressed(os.path.join(HERE, "prompt_encoders.npz"), **out)
    print("prompt_encoders.npz", {k: np.shape(v) for k, v in list(out.items())[:10]}, len(out))


if __name__ == "__main__":
    main()
