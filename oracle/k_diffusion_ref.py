"""Oracle (test infrastructure only): sigma schedule, eps->denoised wrapper, Karras sigmas, DPM++ 2M.

Follows reference `source/modules/external_k_diffusion.py`:
  * `DiscreteSchedule.sigma_to_t / t_to_sigma / get_sigmas` (:58-83)
  * `DiscreteEpsDDPMDenoiser.__init__/get_scalings/forward` (:86-114), `CompVisDenoiser` (:132-139)
Pinned by tests/golden/denoiser.npz.

`get_sigmas_karras` and `sample_dpmpp_2m` live in the un-vendored dependency k_diffusion==0.1.1.post1
(source/requirements.txt; call sites app.py:198, model_k_diffusion.py:143-146,857-859,1175).  They are restated
from the published algorithm (Karras et al. 2022 eq. 5 with rho = 7; Lu et al. 2022 DPM-Solver++(2M)) and are
**parity unpinned**: the reference holds no test or fixture for them.  Self-consistency is tested instead
(known-answer sigmas from SURVEY.md Appendix C, order-2 convergence on an analytic denoiser).
"""
import math

import torch


def sd15_alphas_cumprod():
    """runwayml/stable-diffusion-v1-5 scheduler config: scaled_linear betas 0.00085 -> 0.012, 1000 steps."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class DiscreteEpsDenoiser:
    def __init__(self, alphas_cumprod, quantize=False):
        self.sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5        # :91
        self.log_sigmas = self.sigmas.log()                                  # :47
        self.quantize = quantize
        self.sigma_data = 1.0

    def get_sigmas(self, n=None):                                            # :58-63
        if n is None:
            return torch.cat([self.sigmas.flip(0), self.sigmas.new_zeros([1])])
        t_max = len(self.sigmas) - 1
        t = torch.linspace(t_max, 0, n)
        return torch.cat([self.t_to_sigma(t), self.sigmas.new_zeros([1])])

    def sigma_to_t(self, sigma, quantize=None):                              # :65-77
        quantize = self.quantize if quantize is None else quantize
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        if quantize:
            return dists.abs().argmin(dim=0).view(sigma.shape)
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=self.log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = self.log_sigmas[low_idx], self.log_sigmas[high_idx]
        w = ((low - log_sigma) / (low - high)).clamp(0, 1)
        return ((1 - w) * low_idx + w * high_idx).view(sigma.shape)

    def t_to_sigma(self, t):                                                 # :79-83
        t = t.float()
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[low_idx] + w * self.log_sigmas[high_idx]).exp()

    def get_scalings(self, sigma):                                           # :95-98
        return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

    def forward(self, eps_fn, x, sigma, **kw):                               # :109-114
        c_out, c_in = [s[(...,) + (None,) * (x.ndim - s.ndim)] for s in self.get_scalings(sigma)]
        eps = eps_fn(x * c_in, self.sigma_to_t(sigma), **kw)
        return x[:, :eps.shape[1], ...] + eps * c_out


class DiscreteVDenoiser(DiscreteEpsDenoiser):
    """external_k_diffusion.py:142-172: the model output is v; pinned by tests/golden/vdenoiser.npz"""

    def get_scalings(self, sigma):                                           # :150-154
        sd2 = self.sigma_data ** 2
        return sd2 / (sigma ** 2 + sd2), -sigma * self.sigma_data / (sigma ** 2 + sd2) ** 0.5, 1 / (sigma ** 2 + sd2) ** 0.5

    def forward(self, v_fn, x, sigma, **kw):                                 # :166-171
        c_skip, c_out, c_in = [s[(...,) + (None,) * (x.ndim - s.ndim)] for s in self.get_scalings(sigma)]
        v = v_fn(x * c_in, self.sigma_to_t(sigma), **kw) * c_out
        return v + x[:, :v.shape[1], ...] * c_skip


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    """[parity unpinned] k_diffusion.sampling.get_sigmas_karras: n sigmas + trailing 0, fp32."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho, max_inv_rho = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sig = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sig, sig.new_zeros([1])])


def dpmpp_2m_coeffs(sigmas):
    """Per-step scalars of DPM++ 2M: x <- a*x + b*D_i + c*D_{i-1}.  sigmas: python floats, last == 0."""
    out = []
    for i in range(len(sigmas) - 1):
        s, s_next = sigmas[i], sigmas[i + 1]
        t = -math.log(s)
        if s_next == 0:
            out.append((0.0, 1.0, 0.0))                 # sigma_fn(inf)/sigma = 0, -expm1(-inf) = 1
            continue
        t_next = -math.log(s_next)
        h = t_next - t
        a, e = s_next / s, -math.expm1(-h)
        if i == 0:
            out.append((a, e, 0.0))
        else:
            r = (t + math.log(sigmas[i - 1])) / h       # h_last / h
            out.append((a, e * (1 + 1 / (2 * r)), -e / (2 * r)))
    return out


def sample_dpmpp_2m(model_fn, x, sigmas):
    """[parity unpinned] k_diffusion.sampling.sample_dpmpp_2m.  model_fn(x, sigma[B]) -> denoised."""
    sig = [float(s) for s in sigmas]
    old = None
    for i, (a, b, c) in enumerate(dpmpp_2m_coeffs(sig)):
        den = model_fn(x, sigmas[i] * x.new_ones([x.shape[0]]))
        x = a * x + b * den + (c * old if old is not None and c != 0.0 else 0.0)
        old = den
    return x


def cfg_combine(noise_pred, guidance_scale):
    """model_k_diffusion.py:1162-1166."""
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)
