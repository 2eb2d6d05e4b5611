#!/usr/bin/env python3
"""Golden vectors for the attention-mask branches of the region path, captured by running the REFERENCE's own
`scaled_dot_product_attention_regionstate` (attention_modify.py:74-103) and `AttnProcessor` / `AttnProcessor2_0`
(:106-207, :405-503) in the build container (same stand-in registry as make_golden.py; /root/reference is not present
on the GPU box).  Writes tests/golden/attention_masks.npz - data only:

  a1/ls, a1/s1      float masks that broadcast into [L, S]: `attn_bias += attn_mask` (:89), std over the masked scores
  a1/bool           a bool mask: `attn_mask.masked_fill_(~attn_mask, -inf)` (:86-87) rewrites the MASK (every element
                    True afterwards) and nothing reaches the scores: output == the unmasked output
  a1/b4_raises      a 4-D float mask: torch refuses the in-place add into the [L, S] bias (RuntimeError)
  p1/cross_region_mask   AttnProcessor with attention_mask [B*H, 1, S]: baddbmm(mask, q, k^T, beta=1) then std (:144,166)
  p2/raises         AttnProcessor2_0 with a mask and a region table: views the mask [B, H, 1, S] (:448-452) -> the same
                    RuntimeError;  p2/cross_noregion_mask: without a table it is plain masked SDPA (:483-485)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from inputs import attn_inputs, mask_inputs, proc_inputs  # noqa: E402
from make_golden import DuckAttn, load_ref, ref_weight_func, register_standins  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    register_standins()
    am = load_ref("attention_modify.py")
    wf = ref_weight_func()
    out = {}
    x = attn_inputs("L256_d160", Bc=2, H=8, L=256, S=77, d=160)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    m = mask_inputs()
    sig = torch.tensor(3.25)
    call = lambda mask: am.scaled_dot_product_attention_regionstate(q, k, v, attn_mask=mask, weight_func=wf,  # noqa: E731
                                                                    region_state=w, sigma=sig)
    plain = call(None)
    for name in ("ls", "s1"):
        o = call(torch.from_numpy(m[name]).clone())
        out[f"a1/{name}"] = o[:, :, x["rows"], :].numpy()
        assert (o - plain).abs().max() > 1e-3
    mb = torch.from_numpy(m["bool"]).clone()
    o = call(mb)
    out["a1/bool"] = o[:, :, x["rows"], :].numpy()
    out["a1/bool_equals_unmasked"] = np.bool_(torch.equal(o, plain))
    out["a1/bool_mask_all_true_afterwards"] = np.bool_(bool(mb.all()))
    try:
        call(torch.zeros(2, 8, 1, 77))
        out["a1/b4_raises"] = np.bool_(False)
    except RuntimeError:
        out["a1/b4_raises"] = np.bool_(True)
    out["rows"] = x["rows"]
    # processors
    p = proc_inputs()
    L, S, H = p["L"], p["S"], p["H"]
    hs, enc = torch.from_numpy(p["hidden"]), torch.from_numpy(p["enc"])
    rp = {"region_state": {L: torch.from_numpy(p["w"])}, "sigma": torch.tensor(2.5), "weight_func": wf}
    attn = DuckAttn(p)
    mp = torch.from_numpy(mask_inputs(L=L, S=S, BH=2 * H)["bh1s"])                    # [B*H, 1, S], as prepare_attention_mask returns
    out["p1/cross_region_mask"] = am.AttnProcessor()(attn, hs, encoder_hidden_states=enc, attention_mask=mp.clone(),
                                                     region_prompt=rp).numpy()
    try:
        am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, attention_mask=mp.clone(), region_prompt=rp)
        out["p2/raises"] = np.bool_(False)
    except RuntimeError:
        out["p2/raises"] = np.bool_(True)
    out["p2/cross_noregion_mask"] = am.AttnProcessor2_0()(attn, hs, encoder_hidden_states=enc, attention_mask=mp.clone()).numpy()
    np.savez_compressed(os.path.join(HERE, "attention_masks.npz"), **out)
    print("attention_masks.npz", {k_: (np.shape(v_) if np.ndim(v_) else v_) for k_, v_ in out.items()})


if __name__ == "__main__":
    main()
