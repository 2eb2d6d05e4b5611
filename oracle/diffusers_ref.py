"""Oracle (test infrastructure only): the diffusers-scheduler text-to-image loop of reference
`source/modules/model_diffusers.py` `StableDiffusionPipeline_finetune.__call__` (:340-407) with an Euler discrete scheduler.

diffusers (0.27.2 in the reference's requirements.txt) is absent from /root/reference and from this image: the scheduler
is restated from the published algorithm (Karras et al. 2022, Algorithm 2 without churn; SD1.x scaled-linear betas) -
PARITY UNPINNED.  What IS the reference's own code and is followed line by line: the loop skeleton - CFG duplication
(:345), scale_model_input (:346), region_prompt with sigma = scheduler.sigmas[i] (:349-354), ONE UNet call on all rows with
the std of the scores taken over the WHOLE call (n_std_groups = 1), `u + g (c - u)` (:381-383), scheduler.step
(:390).
"""
import numpy as np
import torch

from . import region_attention as ra
from . import unet_ref


def euler_schedule(num_inference_steps, timestep_spacing="leading", steps_offset=1, num_train_timesteps=1000,
                   beta_start=0.00085, beta_end=0.012):
    """(timesteps fp32 [n], sigmas fp32 [n+1] with trailing 0, init_noise_sigma)"""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    train_sigmas = (((1 - acp) / acp) ** 0.5).numpy()
    n, N = num_inference_steps, num_train_timesteps
    if timestep_spacing == "linspace":
        ts = np.linspace(0, N - 1, n, dtype=np.float32)[::-1].copy()
    elif timestep_spacing == "leading":
        ts = (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.float32) + steps_offset
    else:
        ts = (np.arange(N, 0, -N / n)).round().astype(np.float32) - 1
    sig = np.interp(ts, np.arange(0, N), train_sigmas).astype(np.float32)
    sigmas = np.concatenate([sig, np.zeros(1, dtype=np.float32)])
    init = float(sigmas.max()) if timestep_spacing in ("linspace", "trailing") else float((sigmas.max() ** 2 + 1) ** 0.5)
    return ts, sigmas, init


def euler_txt2img(sd, cfg, latents, text_rows, region_state, guidance_scale, num_inference_steps,
                  timestep_spacing="leading", steps_offset=1):
    """latents: unit-variance noise [n_img,4,h,w] (multiplied by init_noise_sigma here, as diffusers' prepare_latents does);
    text_rows [2*n_img,S,ctx] in the row layout [u.., c..]; returns the final latents (fp32)."""
    ts, sigmas, init = euler_schedule(num_inference_steps, timestep_spacing, steps_offset)
    return euler_run(sd, cfg, latents.float() * init, ts, sigmas, 0, text_rows, region_state, guidance_scale)


def euler_run(sd, cfg, x, ts, sigmas, t_start, text_rows, region_state, guidance_scale, after_step=None, controlnet=None):
    """The loop of reference model_diffusers.py over the timesteps ts[t_start:] (the img2img / inpaint classes truncate the
    schedule by `strength`): the Euler step uses sigmas[t_start + i], the region bias `scheduler.sigmas[i]` - the LOOP index
    into the un-truncated schedule, as the reference writes it.  after_step(i, x) -> x: inpainting's
    re-imposition of the known region; controlnet = {"sd", "cond" [2 n_img rows], "scale": [per step]}: evaluated on the
    scaled input of every step, no region prompt."""
    x = x.float()
    for i, t in enumerate(ts[t_start:]):
        k = t_start + i
        sigma, sigma_next = float(sigmas[k]), float(sigmas[k + 1])
        x_in = torch.cat([x] * 2) / ((sigma ** 2 + 1) ** 0.5)
        tt = torch.full((x_in.shape[0],), float(t))
        rp = {"region_state": region_state, "sigma": float(sigmas[i]), "weight_func": ra.default_weight_func}
        down, mid = None, None
        if controlnet is not None:
            down, mid = unet_ref.controlnet_forward(controlnet["sd"], cfg, x_in, tt, text_rows, controlnet["cond"],
                                                    controlnet["scale"][i])
        eps = unet_ref.unet_forward(sd, cfg, x_in, tt, text_rows, region_prompt=rp, n_std_groups=1, down_residuals=down,
                                    mid_residual=mid)
        u, c = eps.chunk(2)
        eps = u + guidance_scale * (c - u)
        pred_original = x - sigma * eps
        x = x + (x - pred_original) / sigma * (sigma_next - sigma)
        if after_step is not None:
            x = after_step(i, x)
    return x
