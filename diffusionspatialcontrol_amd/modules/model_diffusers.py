"""Diffusers-scheduler text-to-image pipeline - the SECOND caller of the same processor boundary,
counterpart of reference `source/modules/model_diffusers.py` `StableDiffusionPipeline_finetune.__call__` (:136-415).

What it keeps from the reference: the loop skeleton (:340-407) - per scheduler timestep: CFG duplication,
`scheduler.scale_model_input`, `region_prompt = {"region_state", "sigma": scheduler.sigmas[i], "weight_func"}` attached to
`cross_attention_kwargs`, ONE UNet call on [uncond; cond] rows, `u + g (c - u)`, optional rescale, `scheduler.step` - and the
call signature's hot-path arguments.  Differences in SEMANTICS to the k-diffusion pipeline that this path exposes:
  * sigma is the scheduler's fp32 0-dim CPU tensor (:352), not an fp16 device scalar;
  * `num_images_per_prompt > 1` batches images in ONE UNet call and the reference's `qk.std()` then runs over the WHOLE
    batch: this pipeline keeps exactly that (n_std_groups = 1); per-image std groups are the
    k-diffusion pipeline's behaviour.

Schedulers: diffusers is not part of the reference tree nor of this image, so the scheduler protocol
(`set_timesteps / timesteps / sigmas / init_noise_sigma / scale_model_input / step / order`) is implemented by this
module's own `EulerDiscreteScheduler`, restated from the published algorithm (Karras et al. 2022, Alg. 2 without churn =
diffusers 0.27.2 EulerDiscreteScheduler defaults) - PARITY UNPINNED, pinned only by `oracle/diffusers_ref.py` and
self-consistency tests.  Any object with the same protocol works.
Prompt encoding, ControlNet / T2I-Adapter models, the safety checker and latent previews stay 'next' rows.
"""
import math
from typing import Optional

import numpy as np
import torch

from .encode_region_map_function import encode_region_map
from .model_k_diffusion import StableDiffusionPipeline, rescale_noise_cfg


class EulerDiscreteScheduler:
    """scaled-linear betas 0.00085..0.012 over 1000 train steps (SD1.x), epsilon prediction, sigma_t = sqrt((1-acp)/acp),
    inference sigmas by linear interpolation over the (fractional) timestep grid, final sigma 0."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, timestep_spacing="leading",
                 steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.config = type("Cfg", (), {"prediction_type": "epsilon", "num_train_timesteps": num_train_timesteps,
                                       "timestep_spacing": timestep_spacing, "steps_offset": steps_offset})()
        self._train_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.timesteps = self.sigmas = None
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        n, N = num_inference_steps, self.num_train_timesteps
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, N - 1, n, dtype=np.float32)[::-1].copy()
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.float32) + self.steps_offset
        elif self.timestep_spacing == "trailing":
            ts = (np.arange(N, 0, -N / n)).round().astype(np.float32) - 1
        else:
            raise ValueError(self.timestep_spacing)
        sig = np.interp(ts, np.arange(0, N), self._train_sigmas.numpy()).astype(np.float32)
        self.sigmas = torch.from_numpy(np.concatenate([sig, np.zeros(1, dtype=np.float32)]))      # CPU fp32, like diffusers
        self.timesteps = torch.from_numpy(ts).to(device)
        self._step_index = None

    @property
    def init_noise_sigma(self):
        s = self.sigmas.max()
        return s if self.timestep_spacing in ("linspace", "trailing") else (s ** 2 + 1) ** 0.5

    def _index(self, timestep):
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0].item())
        return self._step_index

    def scale_model_input(self, sample, timestep):
        sigma = float(self.sigmas[self._index(timestep)])
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, return_dict=False, **kwargs):
        i = self._index(timestep)
        sigma, sigma_next = float(self.sigmas[i]), float(self.sigmas[i + 1])
        x = sample.float()
        pred_original = x - sigma * model_output.float()
        derivative = (x - pred_original) / sigma
        prev = (x + derivative * (sigma_next - sigma)).to(model_output.dtype)
        self._step_index = i + 1
        return (prev,)


class StableDiffusionPipeline_finetune(StableDiffusionPipeline):
    """`pipe(prompt_embeds=..., negative_prompt_embeds=..., region_map_state=..., ...)` -> [images] (reference :158-415)"""

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 timesteps=None, guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1,
                 eta: float = 0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 ip_adapter_image=None, ip_adapter_image_embeds=None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, cross_attention_kwargs=None, guidance_rescale: float = 0.0, clip_skip=0,
                 region_map_state=None, weight_func=lambda w, sigma, qk: w * sigma * qk.std(), latent_processing=0,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), image_t2i_adapter=None,
                 adapter_conditioning_scale=1.0, adapter_conditioning_factor=1.0, long_encode=0, text_input_ids=None,
                 down_block_additional_residuals=None, mid_block_additional_residual=None, **kwargs):
        if ip_adapter_image is not None or image_t2i_adapter is not None or latent_processing or timesteps is not None:
            raise NotImplementedError("raw IP-Adapter images, T2I-Adapter, latent previews and custom timestep lists are "
                                      "'next' rows (SURVEY.md 8f); pass ip_adapter_image_embeds for IP-Adapter")
        if prompt_embeds is None:
            if prompt is None or self.tokenizer is None or self.text_encoder is None:
                raise NotImplementedError("pass prompt_embeds / negative_prompt_embeds / text_input_ids, or construct the "
                                          "pipeline with a tokenizer and a CLIP text encoder and pass `prompt`")
            from .encoder_prompt_modify import encode_prompt_function
            prompt_embeds, negative_prompt_embeds, text_input_ids = encode_prompt_function(
                self, prompt, self._execution_device, 1, guidance_scale > 1.0, negative_prompt, clip_skip=clip_skip or None,
                long_encode=long_encode)
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        device = self._execution_device
        self._do_classifier_free_guidance = guidance_scale > 1.0
        cfg = self._do_classifier_free_guidance
        n_img = prompt_embeds.shape[0] * num_images_per_prompt
        text = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        if cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds")
            text = torch.cat([negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0), text])
        text = text.to(device=device, dtype=self.unet.dtype)
        added_cond_kwargs = None
        if ip_adapter_image_embeds is not None:                                                       # :278-285, :321-325
            embeds = self.prepare_ip_adapter_image_embeds(None, ip_adapter_image_embeds, device, n_img, cfg)
            added_cond_kwargs = {"image_embeds": [e.to(device=device, dtype=text.dtype) for e in embeds]}
        self.scheduler.set_timesteps(num_inference_steps, device=device)                              # :288
        ts = self.scheduler.timesteps
        if text_input_ids is None:
            text_input_ids = [None, None]
        region_state = encode_region_map(self, region_map_state, width=width, height=height,
                                         num_images_per_prompt=num_images_per_prompt, text_ids=text_input_ids)    # :291-298
        ca_kwargs = {} if cross_attention_kwargs is None else dict(cross_attention_kwargs)
        latents = self.prepare_latents(n_img, self.unet.config.in_channels, height, width, text.dtype, device, generator,
                                       latents)
        latents = latents * float(self.scheduler.init_noise_sigma)                                    # diffusers prepare_latents
        self._text_kv_for(text)
        for i, t in enumerate(ts):
            x_in = torch.cat([latents] * 2) if cfg else latents                                       # :345
            x_in = self.scheduler.scale_model_input(x_in, t)                                          # :346
            ca_kwargs["region_prompt"] = {"region_state": region_state, "sigma": self.scheduler.sigmas[i],
                                          "weight_func": weight_func}                                 # :349-354 (whole-batch std)
            ukw = {} if added_cond_kwargs is None else {"added_cond_kwargs": added_cond_kwargs}
            eps = self.unet(x_in.to(text.dtype), t, encoder_hidden_states=text, cross_attention_kwargs=ca_kwargs,
                            down_block_additional_residuals=down_block_additional_residuals,
                            mid_block_additional_residual=mid_block_additional_residual, return_dict=False, **ukw)[0]
            if cfg:
                u, c = eps.chunk(2)
                eps = u + guidance_scale * (c - u)                                                    # :381-383
                if guidance_rescale > 0.0:
                    eps = rescale_noise_cfg(eps, c, guidance_rescale=guidance_rescale)                # :385-387
            latents = self.scheduler.step(eps, t, latents, return_dict=False)[0]                      # :390
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
        self._drop_text_kv()
        return [self.latent_to_image(latents, output_type)]

    def _text_kv_for(self, text):
        """project (and pack) the text K/V of every cross-attention layer once for the whole loop (the text does not
        change between steps); the k-diffusion pipeline's helper works on its static buffers, here on the plain tensor"""
        self._refresh_text_kv(text)
