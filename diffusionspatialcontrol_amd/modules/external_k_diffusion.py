"""Denoiser wrappers for the k-diffusion sampling path: eps-prediction and v-prediction.

Counterpart of reference `source/modules/external_k_diffusion.py`: `DiscreteSchedule` (:40-83), `DiscreteEpsDDPMDenoiser`
(:86-114), `CompVisDenoiser` (:132-139) - the SD1.5 path - and the v-prediction pair `DiscreteVDDPMDenoiser` /
`CompVisVDenoiser` (:142-182; protocol mode only, pinned by tests/golden/vdenoiser.npz).  Class and method names are the reference's so `model_k_diffusion.StableDiffusionPipeline.setup_unet`
(:138-146) reads the same; the implementation is the build's own:

  * the sigma <-> t interpolation is a binary search over the 1000-entry log-sigma table (`torch.searchsorted`)
    instead of a [1000, B] distance matrix + cumsum + argmax (:67-77) - identical results, pinned by
    tests/golden/denoiser.npz including out-of-range sigmas;
  * `step_scalars(sigma)` gives the per-step host floats (c_in, c_out, t) the fused pipeline needs, computed once
    per schedule in fp64, so the hot loop performs no device reduction and no host sync for them.
"""
import bisect
import math

import torch
from torch import nn

from .sampling import append_dims, append_zero


class DiscreteSchedule(nn.Module):
    def __init__(self, sigmas, quantize):
        super().__init__()
        self.register_buffer("sigmas", sigmas)
        self.register_buffer("log_sigmas", sigmas.log())
        self.quantize = quantize
        self._host_log = None

    sigma_min = property(lambda self: self.sigmas[0])
    sigma_max = property(lambda self: self.sigmas[-1])

    def get_sigmas(self, n=None):
        if n is None:
            return append_zero(self.sigmas.flip(0))
        last = self.sigmas.numel() - 1
        return append_zero(self.t_to_sigma(torch.linspace(last, 0, n, device=self.sigmas.device)))

    def _bracket(self, log_sigma):
        """index of the table entry at or below log_sigma, clamped so that idx + 1 exists"""
        n_le = torch.searchsorted(self.log_sigmas, log_sigma.reshape(-1), right=True)
        return (n_le - 1).clamp(min=0, max=self.log_sigmas.numel() - 2)

    def sigma_to_t(self, sigma, quantize=None):
        quantize = self.quantize if quantize is None else quantize
        ls = sigma.log()
        if quantize:
            return (ls.reshape(1, -1) - self.log_sigmas[:, None]).abs().argmin(dim=0).view(sigma.shape)
        lo = self._bracket(ls)
        a, b = self.log_sigmas[lo], self.log_sigmas[lo + 1]
        frac = ((a - ls.reshape(-1)) / (a - b)).clamp(0, 1)
        return ((1 - frac) * lo + frac * (lo + 1)).view(sigma.shape)

    def t_to_sigma(self, t):
        t = t.float()
        lo, hi = t.floor().long(), t.ceil().long()
        frac = t - lo
        return torch.exp((1 - frac) * self.log_sigmas[lo] + frac * self.log_sigmas[hi])

    # -- host-side scalars for the fused step loop
    def sigma_to_t_host(self, sigma: float) -> float:
        if self._host_log is None:
            self._host_log = [float(v) for v in self.log_sigmas.double().cpu()]
        tab = self._host_log
        ls = math.log(sigma)
        lo = min(max(bisect.bisect_right(tab, ls) - 1, 0), len(tab) - 2)
        frac = min(max((tab[lo] - ls) / (tab[lo] - tab[lo + 1]), 0.0), 1.0)
        return (1 - frac) * lo + frac * (lo + 1)


class DiscreteEpsDDPMDenoiser(DiscreteSchedule):
    """wraps a model that predicts eps on a discrete DDPM schedule"""

    def __init__(self, model, alphas_cumprod, quantize):
        super().__init__(torch.sqrt((1 - alphas_cumprod) / alphas_cumprod), quantize)
        self.inner_model = model
        self.sigma_data = 1.0

    def get_scalings(self, sigma):
        return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

    def step_scalars(self, sigma: float):
        """(c_in, c_out, t) as python floats for one sigma"""
        return 1.0 / math.sqrt(sigma * sigma + self.sigma_data ** 2), -sigma, self.sigma_to_t_host(sigma)

    def get_eps(self, *args, **kwargs):
        return self.inner_model(*args, **kwargs)

    def forward(self, input, sigma, **kwargs):
        c_out, c_in = (append_dims(s, input.ndim) for s in self.get_scalings(sigma))
        eps = self.get_eps(input * c_in, self.sigma_to_t(sigma), **kwargs)
        return input[:, :eps.shape[1]] + eps * c_out     # channel slice: inpaint/controlnet inputs carry extra channels


class CompVisDenoiser(DiscreteEpsDDPMDenoiser):
    """the denoiser `setup_unet` builds around `ModelWrapper` (model_k_diffusion.py:90-98,138-146)"""

    def __init__(self, model, quantize=False, device="cpu"):
        super().__init__(model, model.alphas_cumprod, quantize=quantize)

    def get_eps(self, *args, **kwargs):
        return self.inner_model.apply_model(*args, **kwargs)


class DiscreteVDDPMDenoiser(DiscreteSchedule):
    """wraps a model that predicts v on a discrete DDPM schedule (reference external_k_diffusion.py:142-172)"""

    def __init__(self, model, alphas_cumprod, quantize):
        super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
        self.inner_model = model
        self.sigma_data = 1.0

    def get_scalings(self, sigma):
        sd2 = self.sigma_data ** 2
        c_skip = sd2 / (sigma ** 2 + sd2)
        c_out = -sigma * self.sigma_data / (sigma ** 2 + sd2) ** 0.5
        c_in = 1 / (sigma ** 2 + sd2) ** 0.5
        return c_skip, c_out, c_in

    def get_v(self, *args, **kwargs):
        return self.inner_model(*args, **kwargs)

    def forward(self, input, sigma, **kwargs):
        c_skip, c_out, c_in = (append_dims(s, input.ndim) for s in self.get_scalings(sigma))
        vout = self.get_v(input * c_in, self.sigma_to_t(sigma), **kwargs) * c_out
        return vout + input[:, :vout.shape[1]] * c_skip      # channel slice as in the eps wrapper (:169-171)


class CompVisVDenoiser(DiscreteVDDPMDenoiser):
    """the denoiser `setup_unet` builds for `prediction_type == "v_prediction"` (model_k_diffusion.py:138-139)"""

    def __init__(self, model, quantize=False, device="cpu"):
        super().__init__(model, model.alphas_cumprod, quantize=quantize)

    # Reference defect kept on purpose (SURVEY.md Appendix E policy: decide and document): `get_v` forwards (x, t, cond)
    # only (:181-182), so `cross_attention_kwargs` - and with it the region prompt - never reaches the UNet of a
    # v-prediction model: the reference's spatial control is inert there.  pass_kwargs = True forwards them.
    pass_kwargs = False

    def get_v(self, x, t, cond, **kwargs):
        if self.pass_kwargs:
            return self.inner_model.apply_model(x, t, cond, **kwargs)
        return self.inner_model.apply_model(x, t, cond)
