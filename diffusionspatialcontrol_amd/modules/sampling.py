"""The slice of the un-vendored dependency `k_diffusion.sampling` / `k_diffusion.utils` that the default
pipeline uses (reference call sites: app.py:198 `('DPM++ 2M Karras', 'sample_dpmpp_2m', {'scheduler': 'karras'})`,
model_k_diffusion.py:143-146,857-859,1175; external_k_diffusion.py:29,60).

k_diffusion==0.1.1.post1 is not present in /root/reference nor installable here; these functions restate its published
algorithms (Karras et al. 2022 sigma schedule, rho = 7; Lu et al. 2022 DPM-Solver++(2M)) and are **parity unpinned**
(tests: known-answer sigmas of SURVEY.md Appendix C and second-order convergence).

The other samplers app.py offers (app.py:170-220: Euler, Euler a, LMS, Heun, DPM2, DPM2 a, DPM++ 2S a, DPM++ SDE,
DPM++ 2M SDE (+ Heun), DPM++ 3M SDE; karras / exponential / polyexponential schedules) are restated below from their
published algorithms (Karras et al. 2022 Algorithms 1-2; Lu et al. 2022 DPM-Solver / DPM-Solver++; the ancestral split of
Song et al. 2021) with k-diffusion's call shape `sampler(model, x, sigmas, extra_args=None, callback=None, disable=None,
...)`, as plain torch on whatever device `x` lives on: they drive the same `model_fn` (one UNet graph replay per model
call) in protocol mode; only DPM++ 2M has a fused device-side update.  Tests: order of convergence on an analytic
denoiser, ancestral marginals, eta = 0 reductions (tests/test_host_logic.py).

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


# ----------------------------------------------------------------------------------------------- schedules
def get_sigmas_exponential(n, sigma_min, sigma_max, device="cpu"):
    """n sigmas equally spaced in log sigma + a trailing zero"""
    sigmas = torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp()
    return append_zero(sigmas).to(device)


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0, device="cpu"):
    """polynomial (degree rho) in log sigma + a trailing zero; rho = 1 is the exponential schedule"""
    ramp = torch.linspace(1, 0, n) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return append_zero(sigmas).to(device)


# ----------------------------------------------------------------------------------------------- helpers
def to_d(x, sigma, denoised):
    """Karras ODE derivative dx/dsigma = (x - D(x; sigma)) / sigma"""
    return (x - denoised) / append_dims(sigma, x.ndim)


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """(sigma_down, sigma_up): step deterministically to sigma_down, then add sigma_up of fresh noise, so that the marginal
    at sigma_to is kept: sigma_up^2 = eta^2 sigma_to^2 (1 - sigma_to^2 / sigma_from^2)"""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


class BrownianTreeNoiseSampler:
    """Normalised Brownian increments (W(t1) - W(t0)) / sqrt|t1 - t0| of ONE sample path W over t = transform(sigma),
    reproducible from `seed` - what the SDE samplers need when they query overlapping intervals (DPM++ SDE asks for
    [t, s] and then [t, t_next]: the second increment must contain the first).

    k-diffusion builds this on torchsde's BrownianTree (absent from this image and from the reference tree).  Here the path
    is grown lazily: a new time between two known ones is drawn from the Brownian bridge between them, one outside the
    known range by an independent increment; known times are reused.  Same law as the tree; the VALUES for a given seed
    differ from torchsde's and depend on the order of the queries (parity unpinned, like every sampler here)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda t: t):
        self.shape, self.dtype, self.device = x.shape, x.dtype, x.device
        self.transform = transform
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(0 if seed is None else int(seed))
        t0, t1 = sorted((float(transform(torch.as_tensor(float(sigma_min)))), float(transform(torch.as_tensor(float(sigma_max))))))
        self.ts = [t0, t1]
        self.ws = [torch.zeros(self.shape, dtype=torch.float64), self._randn() * math.sqrt(max(t1 - t0, 0.0))]

    def _randn(self):
        return torch.randn(self.shape, generator=self.gen, dtype=torch.float64)

    def _w(self, t):
        import bisect
        k = bisect.bisect_left(self.ts, t)
        if k < len(self.ts) and self.ts[k] == t:
            return self.ws[k]
        if k == 0:
            w = self.ws[0] - self._randn() * math.sqrt(self.ts[0] - t)
        elif k == len(self.ts):
            w = self.ws[-1] + self._randn() * math.sqrt(t - self.ts[-1])
        else:
            a, b = self.ts[k - 1], self.ts[k]
            lam = (t - a) / (b - a)
            w = self.ws[k - 1] * (1 - lam) + self.ws[k] * lam + self._randn() * math.sqrt((t - a) * (b - t) / (b - a))
        self.ts.insert(k, t)
        self.ws.insert(k, w)
        return w

    def __call__(self, sigma, sigma_next):
        t0 = float(self.transform(torch.as_tensor(float(sigma))))
        t1 = float(self.transform(torch.as_tensor(float(sigma_next))))
        if t0 == t1:
            return torch.zeros(self.shape, dtype=self.dtype, device=self.device)
        inc = (self._w(t1) - self._w(t0)) / math.sqrt(abs(t1 - t0))
        return inc.to(device=self.device, dtype=self.dtype)


def _steps(sigmas):
    """host copies of the schedule: the loop's control flow (`sigma_next == 0`) never syncs on a device tensor"""
    return [float(v) for v in sigmas.detach().cpu().double().tolist()]


def _report(callback, x, i, sigma, sigma_hat, denoised):
    if callback is not None:
        callback({"x": x, "i": i, "sigma": sigma, "sigma_hat": sigma_hat, "denoised": denoised})


# ----------------------------------------------------------------------------------------------- ODE samplers
@torch.no_grad()
def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
    """Karras et al. Algorithm 2 without the second-order correction (Euler steps, optional churn)"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        gamma = min(s_churn / (len(sig) - 1), 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.0
        sigma_hat = sig[i] * (gamma + 1)
        if gamma > 0:
            x = x + torch.randn_like(x) * s_noise * (sigma_hat ** 2 - sig[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        _report(callback, x, i, sigmas[i], sigma_hat, denoised)
        x = x + (x - denoised) * ((sig[i + 1] - sigma_hat) / sigma_hat)
    return x


@torch.no_grad()
def sample_heun(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                s_tmax=float("inf"), s_noise=1.0):
    """Karras et al. Algorithm 2: Euler predictor + trapezoidal corrector (plain Euler on the step to sigma = 0)"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        gamma = min(s_churn / (len(sig) - 1), 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.0
        sigma_hat = sig[i] * (gamma + 1)
        if gamma > 0:
            x = x + torch.randn_like(x) * s_noise * (sigma_hat ** 2 - sig[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = (x - denoised) / sigma_hat
        _report(callback, x, i, sigmas[i], sigma_hat, denoised)
        dt = sig[i + 1] - sigma_hat
        if sig[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            d_2 = (x_2 - model(x_2, sig[i + 1] * s_in, **extra_args)) / sig[i + 1]
            x = x + (d + d_2) * (dt / 2)
    return x


@torch.no_grad()
def sample_dpm_2(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
    """DPM-Solver-2 flavoured midpoint rule: the midpoint is the geometric mean of the two sigmas"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        gamma = min(s_churn / (len(sig) - 1), 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.0
        sigma_hat = sig[i] * (gamma + 1)
        if gamma > 0:
            x = x + torch.randn_like(x) * s_noise * (sigma_hat ** 2 - sig[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in, **extra_args)
        d = (x - denoised) / sigma_hat
        _report(callback, x, i, sigmas[i], sigma_hat, denoised)
        if sig[i + 1] == 0:
            x = x + d * (sig[i + 1] - sigma_hat)
        else:
            sigma_mid = math.exp(0.5 * (math.log(sigma_hat) + math.log(sig[i + 1])))
            x_2 = x + d * (sigma_mid - sigma_hat)
            d_2 = (x_2 - model(x_2, sigma_mid * s_in, **extra_args)) / sigma_mid
            x = x + d_2 * (sig[i + 1] - sigma_hat)
    return x


def linear_multistep_coeff(order, t, i, j):
    """integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial through t_i, t_{i-1}, ..., t_{i-order+1}"""
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")
    nodes = [t[i - k] for k in range(order)]
    others = [nodes[k] for k in range(order) if k != j]
    # expand prod (tau - others[k]) into monomial coefficients and integrate exactly (no quadrature needed)
    poly = [1.0]
    for r in others:
        poly = [0.0] + poly
        for k in range(len(poly) - 1):
            poly[k] -= r * poly[k + 1]
    denom = 1.0
    for r in others:
        denom *= nodes[j] - r
    a, b = t[i], t[i + 1]
    return sum(c * (b ** (k + 1) - a ** (k + 1)) / (k + 1) for k, c in enumerate(poly)) / denom


@torch.no_grad()
def sample_lms(model, x, sigmas, extra_args=None, callback=None, disable=None, order=4):
    """Linear multistep (Adams-Bashforth in sigma, non-uniform nodes), order ramped up over the first steps"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    ds = []
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        ds.append((x - denoised) / sig[i])
        if len(ds) > order:
            ds.pop(0)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        cur_order = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur_order, sig, i, j) for j in range(cur_order)]
        x = x + sum(c * d for c, d in zip(coeffs, reversed(ds)))
    return x


# ----------------------------------------------------------------------------------------------- ancestral samplers
@torch.no_grad()
def sample_euler_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                           noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        x = x + (x - denoised) * ((sigma_down - sig[i]) / sig[i])
        if sig[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * sigma_up)
    return x


@torch.no_grad()
def sample_dpm_2_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                           noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        d = (x - denoised) / sig[i]
        if sigma_down == 0:
            x = x + d * (sigma_down - sig[i])
        else:
            sigma_mid = math.exp(0.5 * (math.log(sig[i]) + math.log(sigma_down)))
            x_2 = x + d * (sigma_mid - sig[i])
            d_2 = (x_2 - model(x_2, sigma_mid * s_in, **extra_args)) / sigma_mid
            x = x + d_2 * (sigma_down - sig[i])
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * sigma_up)
    return x


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                              noise_sampler=None):
    """DPM-Solver++(2S) in t = -log sigma with the ancestral split"""
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    sig = _steps(sigmas)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sigma_down == 0:
            x = x + (x - denoised) * ((sigma_down - sig[i]) / sig[i])
        else:
            t, t_next = -math.log(sig[i]), -math.log(sigma_down)
            h = t_next - t
            s = t + 0.5 * h
            x_2 = (math.exp(-s) / sig[i]) * x - math.expm1(-0.5 * h) * denoised
            denoised_2 = model(x_2, math.exp(-s) * s_in, **extra_args)
            x = (sigma_down / sig[i]) * x - math.expm1(-h) * denoised_2
        if sig[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * (s_noise * sigma_up)
    return x


# ----------------------------------------------------------------------------------------------- SDE samplers
@torch.no_grad()
def sample_dpmpp_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                     noise_sampler=None, r=0.5):
    """DPM-Solver++ (stochastic), two model calls per step"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(x, min(v for v in sig if v > 0), max(sig))
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sig[i + 1] == 0:
            x = x + (x - denoised) * ((sig[i + 1] - sig[i]) / sig[i])
            continue
        t, t_next = -math.log(sig[i]), -math.log(sig[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        # step 1: to the intermediate time s
        sd, su = get_ancestral_step(math.exp(-t), math.exp(-s), eta)
        s_ = -math.log(sd)
        x_2 = (math.exp(-s_) / math.exp(-t)) * x - math.expm1(t - s_) * denoised
        x_2 = x_2 + noise_sampler(math.exp(-t), math.exp(-s)) * (s_noise * su)
        denoised_2 = model(x_2, math.exp(-s) * s_in, **extra_args)
        # step 2: to t_next with the combined estimate
        sd, su = get_ancestral_step(math.exp(-t), math.exp(-t_next), eta)
        t_next_ = -math.log(sd)
        denoised_d = (1 - fac) * denoised + fac * denoised_2
        x = (math.exp(-t_next_) / math.exp(-t)) * x - math.expm1(t - t_next_) * denoised_d
        x = x + noise_sampler(math.exp(-t), math.exp(-t_next)) * (s_noise * su)
    return x


@torch.no_grad()
def sample_dpmpp_2m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                        noise_sampler=None, solver_type="midpoint"):
    """DPM-Solver++(2M) SDE"""
    if solver_type not in {"heun", "midpoint"}:
        raise ValueError("solver_type must be 'heun' or 'midpoint'")
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(x, min(v for v in sig if v > 0), max(sig))
    s_in = x.new_ones([x.shape[0]])
    old_denoised, h_last = None, None
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sig[i + 1] == 0:
            x = denoised
        else:
            t, s = -math.log(sig[i]), -math.log(sig[i + 1])
            h = s - t
            eta_h = eta * h
            x = (sig[i + 1] / sig[i]) * math.exp(-eta_h) * x - math.expm1(-h - eta_h) * denoised
            if old_denoised is not None:
                r = h_last / h
                if solver_type == "heun":
                    x = x + ((-math.expm1(-h - eta_h)) / (-h - eta_h) + 1) * (1 / r) * (denoised - old_denoised)
                else:
                    x = x + 0.5 * (-math.expm1(-h - eta_h)) * (1 / r) * (denoised - old_denoised)
            if eta:
                x = x + noise_sampler(sig[i], sig[i + 1]) * (sig[i + 1] * math.sqrt(-math.expm1(-2 * eta_h)) * s_noise)
            h_last = h
        old_denoised = denoised
    return x


@torch.no_grad()
def sample_dpmpp_3m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0,
                        noise_sampler=None):
    """DPM-Solver++(3M) SDE"""
    extra_args = {} if extra_args is None else extra_args
    sig = _steps(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(x, min(v for v in sig if v > 0), max(sig))
    s_in = x.new_ones([x.shape[0]])
    d1, d2, h1, h2 = None, None, None, None
    for i in range(len(sig) - 1):
        denoised = model(x, sig[i] * s_in, **extra_args)
        _report(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sig[i + 1] == 0:
            x = denoised
        else:
            t, s = -math.log(sig[i]), -math.log(sig[i + 1])
            h = s - t
            h_eta = h * (eta + 1)
            x = math.exp(-h_eta) * x - math.expm1(-h_eta) * denoised
            if h2 is not None:
                r0, r1 = h1 / h, h2 / h
                d1_0 = (denoised - d1) / r0
                d1_1 = (d1 - d2) / r1
                dd1 = d1_0 + (d1_0 - d1_1) * r0 / (r0 + r1)
                dd2 = (d1_0 - d1_1) / (r0 + r1)
                phi_2 = math.expm1(-h_eta) / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                x = x + phi_2 * dd1 - phi_3 * dd2
            elif h1 is not None:
                r = h1 / h
                phi_2 = math.expm1(-h_eta) / h_eta + 1
                x = x + phi_2 * (denoised - d1) / r
            if eta:
                x = x + noise_sampler(sig[i], sig[i + 1]) * (sig[i + 1] * math.sqrt(-math.expm1(-2 * h * eta)) * s_noise)
            h1, h2 = h, h1
        d1, d2 = denoised, d1
    return x
