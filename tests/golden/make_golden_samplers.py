#!/usr/bin/env python3
"""Capture golden vectors for the four extra samplers by running the REFERENCE's own
source/modules/samplers_extra_k_diffusion.py in the build container (/root/reference is absent on the GPU box):

    python tests/golden/make_golden_samplers.py        # rewrites tests/golden/samplers_extra.npz

The reference file imports three names from the un-vendored k_diffusion.sampling (`default_noise_sampler`, `to_d`,
`get_sigmas_karras`) and uses `k_diffusion.sampling.torch`; stand-ins with their published one-line definitions are
registered before the import (so the goldens pin the reference's control flow and arithmetic GIVEN those three).  The
denoiser is the analytic function of tests/golden/inputs.py; noise comes from torch's global CPU generator seeded per
case, so a restatement that draws the same tensors in the same order reproduces the outputs to rounding.
Only data is written."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from inputs import analytic_denoiser, sampler_cases, sampler_start  # noqa: E402

REF = "/root/reference/source/modules/samplers_extra_k_diffusion.py"


def register_standins():
    def append_dims(x, n):
        return x[(...,) + (None,) * (n - x.ndim)]

    def to_d(x, sigma, denoised):
        return (x - denoised) / append_dims(sigma, x.ndim)

    def default_noise_sampler(x):
        return lambda sigma, sigma_next: torch.randn_like(x)

    def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
        ramp = torch.linspace(0, 1, n)
        lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
        s = (hi + ramp * (lo - hi)) ** rho
        return torch.cat([s, s.new_zeros([1])]).to(device)

    kd = types.ModuleType("k_diffusion")
    ks = types.ModuleType("k_diffusion.sampling")
    ks.__dict__.update(default_noise_sampler=default_noise_sampler, to_d=to_d, get_sigmas_karras=get_sigmas_karras, torch=torch)
    kd.sampling = ks
    sys.modules["k_diffusion"], sys.modules["k_diffusion.sampling"] = kd, ks


def main():
    register_standins()
    spec = importlib.util.spec_from_file_location("ref_samplers_extra", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for name, (fn, n, kwargs, seed) in sampler_cases().items():
        x, sigmas = sampler_start(n)
        calls = []

        def model(xx, sigma, **kw):
            calls.append(float(sigma.reshape(-1)[0]))
            return analytic_denoiser(xx, sigma)
        torch.manual_seed(seed)
        y = getattr(ref, fn)(model, x.clone(), sigmas, disable=True, **kwargs)
        out[name] = y.numpy()
        out[name + "/model_sigmas"] = np.array(calls)
        print(f"{name}: {len(calls)} model calls, |y| max {y.abs().max().item():.4f}")
    np.savez_compressed(os.path.join(HERE, "samplers_extra.npz"), **out)


if __name__ == "__main__":
    main()
