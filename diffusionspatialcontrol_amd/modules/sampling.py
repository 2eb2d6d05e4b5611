"""The slice of the un-vendored dependency `k_diffusion.sampling` / `k_diffusion.utils` that the default
pipeline uses (reference call sites: app.py:198 `('DPM++ 2M Karras', 'sample_dpmpp_2m', {'scheduler': 'karras'})`,
model_k_diffusion.py:143-146,857-859,1175; external_k_diffusion.py:29,60).

k_diffusion==0.1.1.post1 is not present in /root/reference nor installable here; these functions restate its published
algorithms (Karras et al. 2022 sigma schedule, rho = 7; Lu et al. 2022 DPM-Solver++(2M)) and are **parity unpinned**
(tests: known-answer sigmas of SURVEY.md Appendix C and second-order convergence).

`sample_dpmpp_2m` keeps k-diffusion's call shape `sampler(model_fn, x, sigmas=...)` so that
`StableDiffusionPipeline.get_scheduler("sample_dpmpp_2m")` resolves here.  The update itself is ONE HIP launch per
step (dsc_dpmpp2m_step) instead of the 3-5 elementwise launches of the torch version; its scalars are computed on the
host in fp64 from the (already fp16-rounded, model_k_diffusion.py:1027-1029) sigmas, so there is no device->host
sync inside the loop (the reference's sampler tests `sigmas[i + 1] == 0` on a device tensor every step).
"""
import math

import torch

from .. import ops


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """n Karras sigmas + a trailing zero, fp32."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas).to(device)


def dpmpp_2m_coefficients(sigmas):
    """Host scalars (a, b, c) per step with  x <- a*x + b*D_i + c*D_{i-1}  (SURVEY.md Appendix C)."""
    sig = [float(s) for s in sigmas]
    coeffs = []
    for i in range(len(sig) - 1):
        s, s_next = sig[i], sig[i + 1]
        if s_next == 0.0:
            coeffs.append((0.0, 1.0, 0.0))            # sigma_next/sigma = 0, -expm1(-inf) = 1: x <- D_i
            continue
        t, t_next = -math.log(s), -math.log(s_next)
        h = t_next - t
        a, e = s_next / s, -math.expm1(-h)
        if i == 0:
            coeffs.append((a, e, 0.0))
        else:
            r = (t + math.log(sig[i - 1])) / h
            coeffs.append((a, e * (1.0 + 1.0 / (2.0 * r)), -e / (2.0 * r)))
    return coeffs


@torch.no_grad()
def sample_dpmpp_2m(model, x, sigmas, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M).  model(x, sigma[B], **extra_args) -> denoised (the CFG-combined estimate)."""
    extra_args = {} if extra_args is None else extra_args
    sig_host = sigmas.detach().float().cpu().tolist()     # one transfer before the loop
    coeffs = dpmpp_2m_coefficients(sig_host)
    s_in = x.new_ones([x.shape[0]])
    old = None
    for i, (a, b, c) in enumerate(coeffs):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        x = ops.dpmpp2m_update(x, denoised, old, a, b, c)
        old = denoised
    return x
