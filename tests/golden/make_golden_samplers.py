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


def capture_vdenoiser():
    """`DiscreteVDDPMDenoiser` / `CompVisVDenoiser` of the reference's external_k_diffusion.py (:142-182) on the SD1.5 schedule
    -> tests/golden/vdenoiser.npz (scalings on a sigma grid, one forward with a recording inner model)"""
    def append_dims(x, n):
        return x[(...,) + (None,) * (n - x.ndim)]
    sys.modules["k_diffusion"].utils = types.ModuleType("k_diffusion.utils")
    sys.modules["k_diffusion"].utils.append_dims = append_dims
    sys.modules["k_diffusion.utils"] = sys.modules["k_diffusion"].utils
    sys.modules["k_diffusion.sampling"].append_zero = lambda x: torch.cat([x, x.new_zeros([1])])
    spec = importlib.util.spec_from_file_location("ref_ekd", os.path.join(os.path.dirname(REF), "external_k_diffusion.py"))
    ek = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ek)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    calls = []

    class Inner:
        alphas_cumprod = acp

        def apply_model(self, x, t, cond=None, **kw):
            calls.append((x.clone(), t.clone(), sorted(kw)))
            return torch.cos(x * 0.9) * 0.4 - 0.02 * t.reshape(-1, 1, 1, 1) / 1000.0

    den = ek.CompVisVDenoiser(Inner())
    out = {}
    grid = torch.tensor([14.6146, 7.0944, 3.1686, 1.0, 0.2480, 0.029168], dtype=torch.float32)
    out["grid"] = grid.numpy()
    for name, v in zip(("c_skip", "c_out", "c_in"), den.get_scalings(grid)):
        out[name] = v.numpy()
    out["sigmas"] = den.sigmas.numpy()
    rng = np.random.default_rng(78)
    x = torch.from_numpy(rng.standard_normal((2, 6, 8, 8)).astype(np.float32))      # 6 channels in, 4 out: the channel slice
    sig = torch.tensor([2.5])

    class Inner4(Inner):
        def apply_model(self, x, t, cond=None, **kw):
            return Inner.apply_model(self, x, t, cond, **kw)[:, :4]
    den4 = ek.CompVisVDenoiser(Inner4())
    out["fwd_x"], out["fwd_sigma"] = x.numpy(), sig.numpy()
    out["fwd_out"] = den4(x, sig, cond=None, cross_attention_kwargs={"k": 1}).numpy()
    out["fwd_inner_x"], out["fwd_inner_t"] = calls[-1][0].numpy(), calls[-1][1].numpy()
    out["kwargs_reach_the_model"] = np.int64(len(calls[-1][2]))                       # 0: the reference drops them (:181-182)
    np.savez_compressed(os.path.join(HERE, "vdenoiser.npz"), **out)
    print("vdenoiser.npz", {k: np.shape(v) for k, v in out.items()})


def main():
    register_standins()
    capture_vdenoiser()
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
