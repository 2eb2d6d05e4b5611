"""Counterparts of reference `source/modules/samplers_extra_k_diffusion.py` - the four samplers app.py wires in as
callables next to the k-diffusion names (app.py:173-176,203: 'LCM', 'Heun++', 'DDPM', 'Restart'):

    restart_sampler  (:7-73)    Restart sampling (Xu et al. 2023): Heun steps with noise re-injection over a restart interval
    sample_ddpm      (:76-105)  ancestral DDPM step expressed in sigma space
    sample_lcm       (:108-120) LCM: jump to the denoised estimate, re-noise to the next sigma
    sample_heunpp2   (:123-175) Heun++ (2- / 3-stage weighted slopes)

Same names, signatures and call shape (`sampler(model, x, sigmas, extra_args=None, callback=None, disable=None, ...)`);
plain torch on the device `x` lives on.  The control flow reads the schedule from a HOST copy (the reference compares
device scalars every step, one sync each).  Pinned by goldens captured from the reference's own file on an analytic
denoiser (tests/golden/samplers_extra.npz; `to_d` / `default_noise_sampler` / `get_sigmas_karras` come from the
un-vendored k_diffusion there and from modules/sampling.py here, so those three stay parity unpinned).
"""
import torch

from .sampling import default_noise_sampler, get_sigmas_karras


def _host(sigmas):
    """exact host copies (fp16 / fp32 / fp64 values are all exact Python floats): comparisons such as `last < sigma` then
    decide exactly as the reference's promoted tensor comparisons do - including its quirk of a tiny spurious re-noising
    when an fp32 restart interval ends one ulp below the schedule's own value"""
    return [float(v) for v in sigmas.detach().cpu().double().tolist()]


def _restart_plan(sig, restart_list):
    """[(sigma_from, sigma_to)] in execution order: the main schedule with each restart interval's extra steps spliced in
    behind the step that reaches its lower end (reference :49-62)."""
    def nearest(v):
        return min(range(len(sig)), key=lambda k: (abs(sig[k] - v), k))      # argmin, first index on ties

    by_index = {nearest(k): v for k, v in restart_list.items()}
    plan = []
    for i in range(len(sig) - 1):
        plan.append((sig[i], sig[i + 1]))
        if i + 1 in by_index:
            n_steps, times, s_max = by_index[i + 1]
            lo, hi = i + 1, nearest(s_max)
            if hi < lo:
                inner = _host(get_sigmas_karras(n_steps, sig[lo], sig[hi])[:-1])
                # the reference decrements `restart_times` inside the dict entry's local copy: each interval runs
                # `times` times, once
                for _ in range(max(times, 0)):
                    plan.extend(zip(inner[:-1], inner[1:]))
    return plan


@torch.no_grad()
def restart_sampler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_noise=1., restart_list=None):
    extra_args = {} if extra_args is None else extra_args
    s_in = x.new_ones([x.shape[0]])
    steps = sigmas.shape[0] - 1
    if restart_list is None:
        if steps >= 20:
            n_restart, times = (steps // 4, 2) if steps >= 36 else (9, 1)
            sigmas = get_sigmas_karras(steps - n_restart * times, sigmas[-2].item(), sigmas[0].item(), device=sigmas.device)
            restart_list = {0.1: [n_restart + 1, times, 2]}
        else:
            restart_list = {}
    sig = _host(sigmas)
    plan = _restart_plan(sig, restart_list)
    last = None
    for step_id, (s_from, s_to) in enumerate(plan):
        if last is not None and last < s_from:                   # a restart: noise back up to the interval's top
            x = x + torch.randn_like(x) * (s_noise * (s_from ** 2 - last ** 2) ** 0.5)
        denoised = model(x, s_from * s_in, **extra_args)
        d = (x - denoised) / s_from
        if callback is not None:
            callback({'x': x, 'i': step_id, 'sigma': s_to, 'sigma_hat': s_from, 'denoised': denoised})
        dt = s_to - s_from
        if s_to == 0:
            x = x + d * dt                                       # Euler on the last step
        else:
            x_2 = x + d * dt
            d_2 = (x_2 - model(x_2, s_to * s_in, **extra_args)) / s_to
            x = x + (d + d_2) / 2 * dt
        last = s_to
    return x


def DDPMSampler_step(x, sigma, sigma_prev, noise, noise_sampler):
    """One DDPM ancestral step in the variance-preserving frame (x here is x_sigma / sqrt(1 + sigma^2))"""
    acp = 1 / (sigma * sigma + 1)
    acp_prev = 1 / (sigma_prev * sigma_prev + 1)
    alpha = acp / acp_prev
    mu = (1.0 / alpha) ** 0.5 * (x - (1 - alpha) * noise / (1 - acp) ** 0.5)
    if sigma_prev > 0:
        mu = mu + ((1 - alpha) * (1. - acp_prev) / (1. - acp)) ** 0.5 * noise_sampler(sigma, sigma_prev)
    return mu


def generic_step_sampler(model, x, sigmas, extra_args=None, callback=None, disable=None, noise_sampler=None,
                         step_function=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    s_in = x.new_ones([x.shape[0]])
    sig = _host(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        x = step_function(x / (1.0 + sig[i] ** 2.0) ** 0.5, sig[i], sig[i + 1], (x - denoised) / sig[i], noise_sampler)
        if sig[i + 1] != 0:
            x = x * (1.0 + sig[i + 1] ** 2.0) ** 0.5
    return x


@torch.no_grad()
def sample_ddpm(model, x, sigmas, extra_args=None, callback=None, disable=None, noise_sampler=None):
    return generic_step_sampler(model, x, sigmas, extra_args, callback, disable, noise_sampler, DDPMSampler_step)


@torch.no_grad()
def sample_lcm(model, x, sigmas, extra_args=None, callback=None, disable=None, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    s_in = x.new_ones([x.shape[0]])
    sig = _host(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        x = denoised
        if sig[i + 1] > 0:
            x = x + sig[i + 1] * noise_sampler(sigmas[i], sigmas[i + 1])
    return x


@torch.no_grad()
def sample_heunpp2(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0.,
                   s_tmax=float('inf'), s_noise=1.):
    extra_args = {} if extra_args is None else extra_args
    s_in = x.new_ones([x.shape[0]])
    sig = _host(sigmas)
    n, s_end = len(sig) - 1, sig[-1]

    def slope(xx, s):
        return (xx - model(xx, s * s_in, **extra_args)) / s

    for i in range(n):
        gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
        eps = torch.randn_like(x) * s_noise                      # drawn every step, like the reference (RNG stream parity)
        sigma_hat = sig[i] * (gamma + 1)
        if gamma > 0:
            x = x + eps * (sigma_hat ** 2 - sig[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = (x - denoised) / sigma_hat
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = sig[i + 1] - sigma_hat
        if sig[i + 1] == s_end:                                  # last step: Euler
            x = x + d * dt
        elif sig[i + 2] == s_end:                                # next-to-last: two slopes
            d_2 = slope(x + d * dt, sig[i + 1])
            w2 = sig[i + 1] / (2 * sig[0])
            x = x + (d * (1 - w2) + d_2 * w2) * dt
        else:                                                    # three slopes
            x_2 = x + d * dt
            d_2 = slope(x_2, sig[i + 1])
            x_3 = x_2 + d_2 * (sig[i + 2] - sig[i + 1])
            d_3 = slope(x_3, sig[i + 2])
            w2, w3 = sig[i + 1] / (3 * sig[0]), sig[i + 2] / (3 * sig[0])
            x = x + ((1 - w2 - w3) * d + w2 * d_2 + w3 * d_3) * dt
    return x
