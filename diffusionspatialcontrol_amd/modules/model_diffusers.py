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

    def add_noise(self, original_samples, noise, timesteps):
        """diffusers EulerDiscreteScheduler.add_noise: x0 + sigma(t) * noise (sigma looked up by timestep VALUE)"""
        t = timesteps.reshape(-1)[0] if torch.is_tensor(timesteps) else timesteps
        idx = int((self.timesteps.to(torch.float32).cpu() == float(t)).nonzero()[0].item())
        return original_samples + float(self.sigmas[idx]) * noise

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
        fixed = None
        if down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            fixed = {"down_block_additional_residuals": down_block_additional_residuals,
                     "mid_block_additional_residual": mid_block_additional_residual}
        latents = self._denoise(latents, ts, text, region_state, weight_func, guidance_scale, guidance_rescale,
                                added_cond_kwargs, ca_kwargs, callback_on_step_end, fixed_residuals=fixed)
        return [self.latent_to_image(latents, output_type)]

    def _denoise(self, latents, ts, text, region_state, weight_func, guidance_scale, guidance_rescale, added_cond_kwargs,
                 ca_kwargs, callback_on_step_end=None, fixed_residuals=None, control=None, after_step=None):
        """The loop every pipeline of reference model_diffusers.py runs (t2i :340-407; the others repeat it): per scheduler
        timestep - CFG duplication, scale_model_input, region_prompt with `sigma = scheduler.sigmas[i]` (i = LOOP index: the
        img2img / inpaint variants thereby read the UN-truncated schedule, SURVEY.md quirk q5 - kept), optional ControlNet call
        on the scaled input, ONE UNet call, `u + g (c - u)`, optional rescale, scheduler.step, optional post-step hook
        (inpainting's re-imposition of the known region)."""
        cfg = self.do_classifier_free_guidance
        from .. import ops
        n_img = latents.shape[0]
        if (ops.PROTOCOL_GRAPH and cfg and fixed_residuals is None and added_cond_kwargs is None and latents.is_cuda
                and text.dtype == torch.float16 and (control is None or not control["guess_mode"])
                and latents.numel() // n_img % 8 == 0):
            return self._denoise_graph(latents, ts, text, region_state, weight_func, guidance_scale, guidance_rescale,
                                       ca_kwargs, callback_on_step_end, control, after_step)
        self._text_kv_for(text)
        for i, t in enumerate(ts):
            x_in = torch.cat([latents] * 2) if cfg else latents                                       # :345
            x_in = self.scheduler.scale_model_input(x_in, t)                                          # :346
            ca_kwargs["region_prompt"] = {"region_state": region_state, "sigma": self.scheduler.sigmas[i],
                                          "weight_func": weight_func}                                 # :349-354 (whole-batch std)
            ukw = {} if added_cond_kwargs is None else {"added_cond_kwargs": added_cond_kwargs}
            if fixed_residuals is not None:
                ukw.update(fixed_residuals)
            if control is not None:                                                                   # :700-760 of the ControlNet classes
                keep = control["keep"][i]
                if isinstance(keep, list):
                    cond_scale = [c_ * s_ for c_, s_ in zip(control["scale"], keep)]
                else:
                    sc_ = control["scale"]
                    cond_scale = (sc_[0] if isinstance(sc_, list) else sc_) * keep
                down, mid = self.controlnet(x_in.to(text.dtype), t, encoder_hidden_states=text, controlnet_cond=control["image"],
                                            conditioning_scale=cond_scale, guess_mode=control["guess_mode"], return_dict=False)
                ukw.update({"down_block_additional_residuals": down, "mid_block_additional_residual": mid})
            eps = self.unet(x_in.to(text.dtype), t, encoder_hidden_states=text, cross_attention_kwargs=ca_kwargs,
                            return_dict=False, **ukw)[0]
            if cfg:
                u, c = eps.chunk(2)
                eps = u + guidance_scale * (c - u)                                                    # :381-383
                if guidance_rescale > 0.0:
                    eps = rescale_noise_cfg(eps, c, guidance_rescale=guidance_rescale)                # :385-387
            latents = self.scheduler.step(eps, t, latents, return_dict=False)[0]                      # :390
            if after_step is not None:
                latents = after_step(i, t, latents)
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
        self._drop_text_kv()
        return latents

    def _text_kv_for(self, text):
        """project (and pack) the text K/V of every cross-attention layer once for the whole loop (the text does not
        change between steps); the k-diffusion pipeline's helper works on its static buffers, here on the plain tensor"""
        self._refresh_text_kv(text)


    def _denoise_graph(self, latents, ts, text, region_state, weight_func, guidance_scale, guidance_rescale, ca_kwargs,
                       callback_on_step_end, control, after_step):
        """The same loop with the UNet (+ ControlNet) call replayed from the k-diffusion pipeline's captured step graph: static
        input / timestep / sigma buffers are filled per step (`sigma = scheduler.sigmas[i]`, whole-batch std: n_std_groups = 1);
        CFG combine and `scheduler.step` stay the scheduler object's eager arithmetic."""
        n_img = latents.shape[0]
        static_control = None
        if control is not None:
            static_control = [{"kind": "controlnet", "image": control["image"]}]
        levels = tuple(sorted((int(L), tuple(w.shape)) for L, w in region_state.items())) if isinstance(region_state, dict) else None
        ckey = None if control is None else (("cn", id(self.controlnet), tuple(
            tuple(i.shape) for i in (control["image"] if isinstance(control["image"], list) else [control["image"]]))),)
        key = ("diffusers", n_img, tuple(latents.shape), levels, tuple(text.shape), text.dtype,
               self._weight_func_key(weight_func), ckey)
        st = self._static_step(key, n_img, tuple(latents.shape), text, region_state, weight_func, ca_kwargs,
                               control=static_control, n_std_groups=1)
        tab = None
        if st["tadd"] is not None:
            # the captured step reads the ResNets' time-embedding terms from st["tadd"] instead of running the embedding path:
            # every timestep of this loop is known here, so their rows are computed once (UNet.temb_add_table) and row i is
            # broadcast into the buffer before replay i.  (The buffer starts as zeros: a replay without this is timestep-blind.)
            tab = self.unet.temb_add_table(torch.tensor([float(t) for t in ts], dtype=torch.float32, device=latents.device))
        for i, t in enumerate(ts):
            x_in = self.scheduler.scale_model_input(torch.cat([latents] * 2), t)                      # :345-346
            st["x_in"].copy_(x_in.to(text.dtype))
            st["t"].fill_(float(t))
            if tab is not None:
                st["tadd"].copy_(tab[i].expand_as(st["tadd"]))
            st["sigma"].fill_(float(self.scheduler.sigmas[i]))                                        # :349-354, loop index (q5)
            if control is not None:
                keep = control["keep"][i]
                sc_ = control["scale"]
                vals = [c_ * s_ for c_, s_ in zip(sc_, keep)] if isinstance(keep, list) else \
                    [(sc_[0] if isinstance(sc_, list) else sc_) * keep]
                for buf, v in zip(st["cn"]["scale"], vals):
                    buf.fill_(float(v))
            st["run"]()
            u, c = st["eps"].chunk(2)
            eps = u + guidance_scale * (c - u)                                                        # :381-383
            if guidance_rescale > 0.0:
                eps = rescale_noise_cfg(eps, c, guidance_rescale=guidance_rescale)                    # :385-387
            latents = self.scheduler.step(eps, t, latents, return_dict=False)[0]                      # :390
            if after_step is not None:
                latents = after_step(i, t, latents)
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
        return latents

    # ---- pieces shared by the other five classes
    def _prepare_call(self, prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, text_input_ids, guidance_scale,
                      num_images_per_prompt, clip_skip, long_encode, ip_adapter_image_embeds, region_map_state, width, height,
                      cross_attention_kwargs):
        device = self._execution_device
        self._do_classifier_free_guidance = guidance_scale > 1.0
        text, text_input_ids, n_img = self._encode_text_rows(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds,
                                                             text_input_ids, num_images_per_prompt, clip_skip or None,
                                                             long_encode, device)
        added = None
        if ip_adapter_image_embeds is not None:
            embeds = self.prepare_ip_adapter_image_embeds(None, ip_adapter_image_embeds, device, n_img,
                                                          self.do_classifier_free_guidance)
            added = {"image_embeds": [e.to(device=device, dtype=text.dtype) for e in embeds]}
        region_state = encode_region_map(self, region_map_state, width=width, height=height,
                                         num_images_per_prompt=num_images_per_prompt, text_ids=text_input_ids)
        return device, text, n_img, added, region_state, ({} if cross_attention_kwargs is None else dict(cross_attention_kwargs))

    def get_timesteps(self, num_inference_steps, strength, device):
        """diffusers img2img `get_timesteps`: keep the last int(n * strength) steps"""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    def _image_latents(self, image, n_img, dtype, device, generator):
        """4-channel tensors are latents already; pixels go through the VAE encoder (posterior sample x scaling factor)"""
        if isinstance(image, torch.Tensor) and image.shape[1] == 4:
            lat = image.to(device=device, dtype=dtype)
        else:
            lat = self._encode_vae_image(self.preprocess(image), generator).to(device=device, dtype=dtype)
        return lat.repeat(n_img // lat.shape[0], 1, 1, 1)

    def _control(self, control_image, controlnet_conditioning_scale, control_guidance_start, control_guidance_end, width,
                 height, n_steps, n_img, num_images_per_prompt):
        if self.controlnet is None:
            raise ValueError("this pipeline needs a ControlNet: construct it with controlnet=... or call setup_controlnet()")
        image, keep, guess_mode, scale = self.preprocess_controlnet(controlnet_conditioning_scale, control_guidance_start,
                                                                    control_guidance_end, control_image, width, height,
                                                                    n_steps, n_img, num_images_per_prompt)
        return {"image": image, "keep": keep, "guess_mode": guess_mode, "scale": scale}


class StableDiffusionImg2ImgPipeline_finetune(StableDiffusionPipeline_finetune):
    """reference model_diffusers.py:1228-1513: encode the image, `scheduler.add_noise` at the first kept timestep, run the
    last `strength` fraction of the schedule."""

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, strength: float = 0.8, num_inference_steps: int = 50, timesteps=None,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0,
                 generator=None, prompt_embeds=None, negative_prompt_embeds=None, ip_adapter_image=None,
                 ip_adapter_image_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs=None, guidance_rescale: float = 0.0, clip_skip=0, region_map_state=None,
                 weight_func=lambda w, sigma, qk: w * sigma * qk.std(), latent_processing=0, callback_on_step_end=None,
                 image_t2i_adapter=None, long_encode=0, text_input_ids=None, height=None, width=None,
                 control_image=None, controlnet_conditioning_scale=1.0, control_guidance_start=0.0, control_guidance_end=1.0,
                 **kwargs):
        if ip_adapter_image is not None or image_t2i_adapter is not None or latent_processing or timesteps is not None:
            raise NotImplementedError("raw IP-Adapter images, T2I-Adapter, latent previews and custom timestep lists are "
                                      "'next' rows (SURVEY.md 8f)")
        if image is None:
            raise ValueError("img2img needs `image` (pixels in [-1, 1] / PIL, or 4-channel latents)")
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if height is None or width is None:
            hh, ww = (image.shape[-2:] if isinstance(image, torch.Tensor) else image.size[::-1])
            f = 8 if (isinstance(image, torch.Tensor) and image.shape[1] == 4) else 1
            height, width = height or int(hh) * f, width or int(ww) * f
        device, text, n_img, added, region_state, ca = self._prepare_call(
            prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, text_input_ids, guidance_scale,
            num_images_per_prompt, clip_skip, long_encode, ip_adapter_image_embeds, region_map_state, width, height,
            cross_attention_kwargs)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        ts, _ = self.get_timesteps(num_inference_steps, strength, device)
        if len(ts) < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                             "pipeline steps is 0 which is < 1 and not appropriate for this pipeline.")
        init = self._image_latents(image, n_img, text.dtype, device, generator)
        noise = self._randn_like_ref(init.shape, generator, device, text.dtype)
        latents = self.scheduler.add_noise(init, noise, ts[:1])
        control = None
        if control_image is not None or self._needs_control:
            control = self._control(control_image, controlnet_conditioning_scale, control_guidance_start, control_guidance_end,
                                    width, height, len(ts), n_img, num_images_per_prompt)
        latents = self._denoise(latents, ts, text, region_state, weight_func, guidance_scale, guidance_rescale, added, ca,
                                callback_on_step_end, control=control)
        return [self.latent_to_image(latents, output_type)]

    _needs_control = False


StableDiffusionPipeline_finetune._needs_control = False


class StableDiffusionInpaintPipeline_finetune(StableDiffusionPipeline_finetune):
    """reference model_diffusers.py:1515-1918, the 4-channel UNet branch of diffusers' inpainting loop: after every scheduler
    step the known region is re-imposed - `(1 - mask) * add_noise(image_latents, noise, t_next) + mask * latents`."""

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, mask_image=None, masked_image_latents=None, height=None, width=None,
                 padding_mask_crop=None, strength: float = 1.0, num_inference_steps: int = 50, timesteps=None,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, ip_adapter_image=None,
                 ip_adapter_image_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs=None, guidance_rescale: float = 0.0, clip_skip=0, region_map_state=None,
                 weight_func=lambda w, sigma, qk: w * sigma * qk.std(), latent_processing=0, callback_on_step_end=None,
                 image_t2i_adapter=None, long_encode=0, text_input_ids=None,
                 control_image=None, controlnet_conditioning_scale=1.0, control_guidance_start=0.0, control_guidance_end=1.0,
                 **kwargs):
        if ip_adapter_image is not None or image_t2i_adapter is not None or latent_processing or timesteps is not None \
                or padding_mask_crop is not None:
            raise NotImplementedError("raw IP-Adapter images, T2I-Adapter, latent previews, custom timestep lists and mask "
                                      "cropping are 'next' rows (SURVEY.md 8f)")
        if self.unet.config.in_channels != 4:
            raise NotImplementedError("the 9-channel inpainting UNet is served by the k-diffusion pipeline's `inpaiting`; this "
                                      "diffusers-scheduler class implements the 4-channel branch")
        if image is None or mask_image is None:
            raise ValueError("inpainting needs `image` and `mask_image`")
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        device, text, n_img, added, region_state, ca = self._prepare_call(
            prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, text_input_ids, guidance_scale,
            num_images_per_prompt, clip_skip, long_encode, ip_adapter_image_embeds, region_map_state, width, height,
            cross_attention_kwargs)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        ts, _ = self.get_timesteps(num_inference_steps, strength, device)
        if len(ts) < 1:
            raise ValueError("After adjusting the num_inference_steps by strength parameter, the number of pipeline steps is 0")
        is_strength_max = strength == 1.0
        init_image = image if (isinstance(image, torch.Tensor) and image.shape[1] == 4) else self._image_tensor(image, height, width)
        image_latents = self._image_latents(init_image, n_img, text.dtype, device, generator)
        noise = self._randn_like_ref(image_latents.shape, generator, device, text.dtype) if latents is None \
            else latents.to(device=device, dtype=text.dtype)
        if is_strength_max or latents is not None:          # diffusers prepare_latents: passed latents are pure noise
            x = noise * float(self.scheduler.init_noise_sigma)
        else:
            x = self.scheduler.add_noise(image_latents, noise, ts[:1])
        mask = self._image_tensor(mask_image, height, width, mask=True)
        mask = torch.nn.functional.interpolate(mask, size=(height // self.vae_scale_factor, width // self.vae_scale_factor))
        mask = mask.to(device=device, dtype=text.dtype)
        if mask.shape[0] < n_img:
            mask = mask.repeat(n_img // mask.shape[0], 1, 1, 1)

        def keep_known(i, t, lat):
            known = image_latents
            if i < len(ts) - 1:
                known = self.scheduler.add_noise(known, noise, ts[i + 1:i + 2])
            return (1 - mask) * known + mask * lat

        control = None
        if control_image is not None or self._needs_control:
            control = self._control(control_image, controlnet_conditioning_scale, control_guidance_start, control_guidance_end,
                                    width, height, len(ts), n_img, num_images_per_prompt)
        x = self._denoise(x, ts, text, region_state, weight_func, guidance_scale, guidance_rescale, added, ca,
                          callback_on_step_end, control=control, after_step=keep_known)
        return [self.latent_to_image(x, output_type)]


class _WithControlNet:
    """constructor of the three ControlNet classes: diffusers puts `controlnet` between `unet` and `scheduler`"""
    _needs_control = True

    def __init__(self, vae, text_encoder, tokenizer, unet, controlnet, scheduler, feature_extractor=None, image_encoder=None):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler, feature_extractor, image_encoder)
        self.setup_controlnet(controlnet)


class StableDiffusionControlNetPipeline_finetune(_WithControlNet, StableDiffusionPipeline_finetune):
    """reference model_diffusers.py:418-823: the t2i loop with a ControlNet evaluated on the scaled input of every step"""

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, height=None, width=None, num_inference_steps: int = 50, timesteps=None,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, ip_adapter_image=None,
                 ip_adapter_image_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs=None, controlnet_conditioning_scale=1.0, guess_mode: bool = False,
                 control_guidance_start=0.0, control_guidance_end=1.0, guidance_rescale: float = 0.0, clip_skip=0,
                 region_map_state=None, weight_func=lambda w, sigma, qk: w * sigma * qk.std(), latent_processing=0,
                 callback_on_step_end=None, image_t2i_adapter=None, long_encode=0, text_input_ids=None, **kwargs):
        if ip_adapter_image is not None or image_t2i_adapter is not None or latent_processing or timesteps is not None:
            raise NotImplementedError("raw IP-Adapter images, T2I-Adapter, latent previews and custom timestep lists are "
                                      "'next' rows (SURVEY.md 8f)")
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        device, text, n_img, added, region_state, ca = self._prepare_call(
            prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, text_input_ids, guidance_scale,
            num_images_per_prompt, clip_skip, long_encode, ip_adapter_image_embeds, region_map_state, width, height,
            cross_attention_kwargs)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        ts = self.scheduler.timesteps
        control = self._control(image, controlnet_conditioning_scale, control_guidance_start, control_guidance_end, width,
                                height, len(ts), n_img, num_images_per_prompt)
        x = self.prepare_latents(n_img, self.unet.config.in_channels, height, width, text.dtype, device, generator, latents)
        x = x * float(self.scheduler.init_noise_sigma)
        x = self._denoise(x, ts, text, region_state, weight_func, guidance_scale, guidance_rescale, added, ca,
                          callback_on_step_end, control=control)
        return [self.latent_to_image(x, output_type)]


class StableDiffusionControlNetImg2ImgPipeline_finetune(_WithControlNet, StableDiffusionImg2ImgPipeline_finetune):
    """reference model_diffusers.py:826-1225: img2img + ControlNet (`image` = the picture, `control_image` = the condition)"""


class StableDiffusionControlNetInpaintPipeline_finetune(_WithControlNet, StableDiffusionInpaintPipeline_finetune):
    """reference model_diffusers.py:1921-: inpainting + ControlNet"""
