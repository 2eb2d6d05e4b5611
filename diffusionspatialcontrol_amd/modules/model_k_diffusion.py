"""k-diffusion text-to-image pipeline for the MI355X hot path - counterpart of reference
`source/modules/model_k_diffusion.py` (`ModelWrapper` :85-98, `StableDiffusionPipeline` :101-, `setup_unet`
:138-141, `get_scheduler` :143-146, `get_sigmas` :848-882, `prepare_latents` :428-456, `txt2img` :943-1231).

Scope (SURVEY.md 8a rows a6-a10, 8f and the callers around them): `txt2img` (:943-1231), `img2img` (:543-846), `inpaiting`
(:1365-1760; 4- and 9-channel UNets), hires upscale, ControlNet (`setup_controlnet` / `preprocess_controlnet`, :348-427),
T2I-Adapter, IP-Adapter (embeddings or raw images through `encode_image`), prompt strings through
`encoder_prompt_modify.encode_prompt_function` (all three `long_encode` branches), every sampler `app.py:170-220` lists, eps-
and v-prediction.  Latent previews (`latent_processing == 1`) return the list of per-model-call estimates as the reference does.
Not built (NotImplementedError): mask cropping, LoRA scaling of the text encoder, textual inversion.  `output_type="latent"` (an extra of this build) returns the final latents.

Two execution modes produce the same numbers:
  * protocol mode (`fused=False`): the reference's control flow - a `model_fn(x, sigma)` closure handed to
    `sampler(model_fn, latents, sigmas=...)` (:1175), any sampler callable.  The closure's body is graph-backed where nothing
    but (x, sigma) changes per call: one launch writes the static inputs, ONE replay of the captured UNet (+ ControlNet /
    adapter) step, one launch does CFG + eps -> denoised; otherwise (v-prediction, IP-Adapter embeddings, CFG rescale, 9-channel
    inpainting) it is the reference's eager sequence - duplicate the latent, attach `region_prompt`, `CompVisDenoiser` ->
    `ModelWrapper.apply_model` -> UNet, combine CFG (:1091-1171).
  * fused mode (`fused=True`, default for `sample_dpmpp_2m`): the UNet forward of one step is captured ONCE into a
    HIP graph over static buffers (sigma and the timestep are device scalars, so the same graph serves all 25
    steps); between replays ONE HIP launch (dsc_cfg_dpmpp2m_step) does CFG combine + eps->denoised + the DPM++ 2M
    update + the next step's `cat([x]*2) * c_in`.  No device->host sync inside the loop (the reference has two
    per step: `sigma.item()` :1115 and the sampler's `sigmas[i+1] == 0` test).
Batches of B > 1 images use the reference's row layout [u_0..u_{B-1}, c_0..c_{B-1}] (:1021,1097) with one std group
per image (`n_std_groups = B`), which is what B separate reference calls compute (the reference itself cannot batch:
external_k_diffusion.py:109-114 broadcasts c_in[B] against input[2B]).
"""
import importlib
import inspect
import os
import threading
import time
from typing import List, Optional, Union

import torch

from .. import _lib, ops
from . import sampling
from .encode_region_map_function import encode_region_map
from .attention_modify import weight_func_is_default
from .external_k_diffusion import CompVisDenoiser, CompVisVDenoiser


class ModelWrapper:
    def __init__(self, model, alphas_cumprod):
        self.model = model
        self.alphas_cumprod = alphas_cumprod

    def apply_model(self, *args, **kwargs):
        if len(args) == 3:
            encoder_hidden_states = args[-1]
            args = args[:2]
        if kwargs.get("cond", None) is not None:
            encoder_hidden_states = kwargs.pop("cond")
        return self.model(*args, encoder_hidden_states=encoder_hidden_states, **kwargs).sample


class SD15Scheduler:
    """Stand-in for the diffusers scheduler object the reference pipeline reads `alphas_cumprod` and
    `config.prediction_type` from (:138-141): scaled_linear betas 0.00085 -> 0.012, 1000 steps (SD1.5)."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000, prediction_type="epsilon"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = type("Cfg", (), {"prediction_type": prediction_type})()


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """reference :71-82 (arXiv 2305.08891 sec. 3.4)"""
    dims = list(range(1, noise_pred_text.ndim))
    factor = noise_pred_text.std(dim=dims, keepdim=True) / noise_cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (noise_cfg * factor) + (1 - guidance_rescale) * noise_cfg


_CAPTURE_LOCK = threading.RLock()


class StableDiffusionPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, feature_extractor=None, image_encoder=None):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler = unet, scheduler
        self.feature_extractor, self.image_encoder = feature_extractor, image_encoder
        self.controlnet = None
        self.adapter = None                                  # T2I-Adapter, set by modules.t2i_adapter.setup_model_t2i_adapter
        self.vae_scale_factor = 8 if vae is None else 2 ** (len(vae.config.block_out_channels) - 1)
        self._do_classifier_free_guidance = True
        self._graphs = {}
        self._added_cond_kwargs = None
        self.setup_unet(self.unet)

    # ------------------------------------------------------------------ reference surface
    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.unet.device

    @property
    def do_classifier_free_guidance(self):
        return self._do_classifier_free_guidance

    def to(self, device):
        self.unet.to(device)
        self.k_diffusion_model.to(device)
        return self

    def setup_unet(self, unet):
        self.unet = unet
        model = ModelWrapper(unet, self.scheduler.alphas_cumprod)
        self.v_prediction = getattr(self.scheduler.config, "prediction_type", "epsilon") == "v_prediction"      # :138-141
        self.k_diffusion_model = CompVisVDenoiser(model) if self.v_prediction else CompVisDenoiser(model)
        self.k_diffusion_model.to(unet.device)
        self._graphs = {}
        self._drop_text_kv()

    def get_scheduler(self, scheduler_type: str):
        """reference :143-146 resolves the name in `k_diffusion.sampling`; here in the build's own sampling module"""
        return getattr(importlib.import_module(sampling.__name__), scheduler_type)

    def setup_controlnet(self, controlnet):
        """reference :348-353"""
        from .controlnet import MultiControlNetModel
        if isinstance(controlnet, (list, tuple)):
            controlnet = MultiControlNetModel(controlnet)
        self.controlnet = controlnet

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype,
                      do_classifier_free_guidance=False, guess_mode=False):
        """reference :483-515: the control image in [0, 1] (no normalisation) at (height, width), duplicated for CFG"""
        if not isinstance(image, torch.Tensor):
            image = (self._image_tensor(image, height, width) + 1.0) / 2.0
        elif tuple(image.shape[-2:]) != (height, width):
            image = torch.nn.functional.interpolate(image.float(), size=(height, width), mode="bilinear")
        image = image.to(device=device, dtype=dtype)
        if do_classifier_free_guidance and not guess_mode:
            image = torch.cat([image] * 2)
        return image

    def preprocess_controlnet(self, controlnet_conditioning_scale, control_guidance_start, control_guidance_end, image, width,
                              height, num_inference_steps, batch_size, num_images_per_prompt):
        """reference :355-427: (control image(s), keep schedule, guess_mode, conditioning scale(s))"""
        from .controlnet import ControlNetModel, MultiControlNetModel
        controlnet = self.controlnet
        multi = isinstance(controlnet, MultiControlNetModel)
        if not isinstance(control_guidance_start, list) and isinstance(control_guidance_end, list):
            control_guidance_start = len(control_guidance_end) * [control_guidance_start]
        elif not isinstance(control_guidance_end, list) and isinstance(control_guidance_start, list):
            control_guidance_end = len(control_guidance_start) * [control_guidance_end]
        elif not isinstance(control_guidance_start, list) and not isinstance(control_guidance_end, list):
            mult = len(controlnet.nets) if multi else 1
            control_guidance_start, control_guidance_end = mult * [control_guidance_start], mult * [control_guidance_end]
        if multi and isinstance(controlnet_conditioning_scale, float):
            controlnet_conditioning_scale = [controlnet_conditioning_scale] * len(controlnet.nets)
        first = controlnet.nets[0] if multi else controlnet
        guess_mode = bool(first.config.global_pool_conditions)
        prep = lambda im: self.prepare_image(im, width, height, batch_size, num_images_per_prompt, self._execution_device,  # noqa: E731
                                             controlnet.dtype, self.do_classifier_free_guidance, guess_mode)
        if multi:
            image = [prep(im) for im in image]
        elif isinstance(controlnet, ControlNetModel):
            image = prep(image)
        else:
            raise TypeError("setup_controlnet() takes a ControlNetModel or a list of them")
        keep = []
        for i in range(num_inference_steps):
            keeps = [1.0 - float(i / num_inference_steps < s or (i + 1) / num_inference_steps > e)
                     for s, e in zip(control_guidance_start, control_guidance_end)]
            keep.append(keeps if multi else keeps[0])
        return image, keep, guess_mode, controlnet_conditioning_scale

    def _controlnet_hook(self, control_img, controlnet_conditioning_scale, control_guidance_start, control_guidance_end,
                         width, height, n_steps, n_img, num_images_per_prompt, text):
        """The per-call ControlNet evaluation of the reference's model_fn (:1118-1152) as a function
        (latent_model_input, sigma) -> UNet keyword arguments; None when no ControlNet is set up."""
        if self.controlnet is None:
            return None
        if control_img is None:
            raise ValueError("a ControlNet is set up (setup_controlnet): pass control_img")
        scale = 1.0 if controlnet_conditioning_scale is None else controlnet_conditioning_scale
        start = 0.0 if control_guidance_start is None else control_guidance_start
        end = 1.0 if control_guidance_end is None else control_guidance_end
        img, keep, guess_mode, scale = self.preprocess_controlnet(scale, start, end, control_img, width, height, n_steps,
                                                                  n_img, num_images_per_prompt)
        seen = []
        kdm = self.k_diffusion_model
        cfg = self.do_classifier_free_guidance

        def schedule(key):
            """conditioning scale(s) of the model call at sigma `key`: one control step per DISTINCT sigma, in call order"""
            if key not in seen:                                 # (:1119-1123)
                seen.append(key)
            k = keep[min(len(seen) - 1, len(keep) - 1)]
            if isinstance(k, list):
                return [c * s_ for c, s_ in zip(scale, k)]
            return (scale[0] if isinstance(scale, list) else scale) * k

        def hook(latent_model_input, sigma):
            cond_scale = schedule(float(sigma[0]))
            down, mid = self.controlnet(latent_model_input / ((sigma[0] ** 2 + 1) ** 0.5), kdm.sigma_to_t(sigma),
                                        encoder_hidden_states=text, controlnet_cond=img, conditioning_scale=cond_scale,
                                        guess_mode=guess_mode, return_dict=False)
            if guess_mode and cfg:                              # inferred for the conditional rows only (:1143-1148)
                down = [torch.cat([torch.zeros_like(d), d]) for d in down]
                mid = torch.cat([torch.zeros_like(mid), mid])
            return {"down_block_additional_residuals": down, "mid_block_additional_residual": mid}
        hook.static = None if guess_mode else {"kind": "controlnet", "image": img, "schedule": schedule}
        return hook

    def _adapter_hook(self, image_t2i_adapter, adapter_conditioning_scale, adapter_conditioning_factor, width, height,
                      steps_denoising, num_images_per_prompt):
        """T2I-Adapter (reference :700-703, :722-731): the adapter's features are computed once; every model call during the
        first int(steps_denoising * factor) distinct sigmas hands clones of them to the UNet as
        `down_intrablock_additional_residuals`.  None when no adapter image is given."""
        if image_t2i_adapter is None:
            return None
        if getattr(self, "adapter", None) is None:
            raise ValueError("image_t2i_adapter needs an adapter: modules.t2i_adapter.setup_model_t2i_adapter(pipe, adapter)")
        from .t2i_adapter import preprocessing_t2i_adapter
        state = preprocessing_t2i_adapter(self, image_t2i_adapter, width, height, adapter_conditioning_scale, num_images_per_prompt)
        seen = []
        limit = int(steps_denoising * adapter_conditioning_factor)

        def schedule(key):
            use = len(seen) < limit
            if key not in seen:
                seen.append(key)
            return use

        def hook(latent_model_input, sigma):
            return {"down_intrablock_additional_residuals": [v.clone() for v in state]} if schedule(float(sigma[0])) else {}
        hook.static = {"kind": "adapter", "state": state, "schedule": schedule}
        return hook

    @staticmethod
    def _merge_hooks(*hooks):
        hooks = [h for h in hooks if h is not None]
        if not hooks:
            return None

        def merged(latent_model_input, sigma):
            out = {}
            for h in hooks:
                out.update(h(latent_model_input, sigma))
            return out
        statics = [getattr(h, "static", None) for h in hooks]
        merged.static = None if any(st_ is None for st_ in statics) else statics
        return merged

    def get_sigmas(self, steps, params):
        """reference :848-882"""
        discard = params.get("discard_next_to_last_sigma", False)
        steps += 1 if discard else 0
        named = {"karras": sampling.get_sigmas_karras, "exponential": sampling.get_sigmas_exponential,
                 "polyexponential": sampling.get_sigmas_polyexponential}
        if params.get("scheduler", None) in named:
            # sigma_min / sigma_max are constants of the model: read once (a device->host copy synchronises, and a
            # synchronisation per generation keeps the host from preparing generation i+1 under generation i's replays);
            # the 26-entry schedule is computed on the host
            kdm = self.k_diffusion_model
            rng = getattr(self, "_sigma_range", None)
            if rng is None or rng[0] is not kdm:
                rng = self._sigma_range = (kdm, kdm.sigmas[0].item(), kdm.sigmas[-1].item())
            sigmas = named[params["scheduler"]](n=steps, sigma_min=rng[1], sigma_max=rng[2], device="cpu")
        else:
            sigmas = self.k_diffusion_model.get_sigmas(steps)
        if discard:
            sigmas = torch.cat([sigmas[:-2], sigmas[-1:]])
        return sigmas

    def _schedule(self, steps, params, device, dtype):
        """get_sigmas() cast to the model dtype on the device (reference :1027-1029: fp16 rounding is part of the schedule).
        A host-computed schedule is rounded on the host (round-to-nearest-even, as the device cast), uploaded from pinned
        memory without blocking, and carries the rounded values as host floats (`_dsc_host`) for the fused loop."""
        s = self.get_sigmas(steps, params)
        device = torch.device(device)
        if s.is_cuda or device.type != "cuda":
            return s.to(device, dtype=dtype)
        h = s.to(dtype=dtype)
        d = h.pin_memory().to(device, non_blocking=True)
        d._dsc_host = h.float().tolist()
        return d

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // 8, width // 8)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device)
        return latents

    def get_sampler_extra_args_t2i(self, sigmas, eta, steps, sampler_opt, latents, seed, func):
        params = inspect.signature(func).parameters
        extra = {}
        if "eta" in params:
            extra["eta"] = eta
        if "sigma_min" in params:
            extra["sigma_min"], extra["sigma_max"] = sigmas[0].item(), sigmas[-1].item()
        if "n" in params:
            extra["n"] = steps
        else:
            extra["sigmas"] = sigmas
        if sampler_opt.get("brownian_noise", False):
            extra["noise_sampler"] = self.create_noise_sampler(latents, sigmas, steps, seed)
        if sampler_opt.get("solver_type", None) == "heun":
            extra["solver_type"] = "heun"
        return extra

    def create_noise_sampler(self, x, sigmas, p, seed):
        """reference :884-890 (BrownianTreeNoiseSampler seeded per generation; see sampling.BrownianTreeNoiseSampler for
        what the torchsde-free counterpart keeps of it)"""
        sigma_min, sigma_max = sigmas[sigmas > 0].min(), sigmas.max()
        return sampling.BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=seed)

    @torch.no_grad()
    def decode_latents(self, latents):
        """reference :291-299: 1/scaling_factor, vae.decode, /2 + 0.5, clamp, NHWC float32 numpy"""
        latents = latents.to(self.device, dtype=self.vae.dtype)
        latents = 1 / self.vae.config.scaling_factor * latents
        image = self.vae.decode(latents).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [Image.fromarray(im) for im in images]

    def latent_to_image(self, latents, output_type):
        """reference :533-539; output_type 'latent' (this build's extra) returns the latents untouched"""
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise NotImplementedError("no VAE was given to the pipeline: pass output_type='latent' or construct it with a "
                                      "modules.vae_decoder.AutoencoderKLDecoder")
        image = self.decode_latents(latents)
        if output_type == "pil":
            image = self.numpy_to_pil(image)
        if len(image) > 1:
            return image
        return image[0]

    # ---- IP-Adapter surface (reference ip_adapter.py:48-292, state-dict form)
    def load_ip_adapter(self, pretrained_model_name_or_path_or_dict, subfolder=None, weight_name=None,
                        image_encoder_folder=None, **kwargs):
        """ip_adapter.py:48-239 restricted to what works offline: `pretrained_model_name_or_path_or_dict` must be a
        state dict {"image_proj": ..., "ip_adapter": ...} or a list of them; the CLIP image encoder is not loaded
        (pass `ip_adapter_image_embeds` to txt2img, as the reference's own warning at :219-222 suggests)."""
        sds = pretrained_model_name_or_path_or_dict
        if not isinstance(sds, list):
            sds = [sds]
        for sd in sds:
            if not isinstance(sd, dict) or sorted(sd.keys()) != ["image_proj", "ip_adapter"]:
                raise ValueError("Required keys are (`image_proj` and `ip_adapter`) missing from the state dict.")   # :191-192
        self._graphs = {}
        return self.unet._load_ip_adapter_weights(sds, low_cpu_mem_usage=False)

    def set_ip_adapter_scale(self, scale):
        """ip_adapter.py:242-264"""
        from .attention_modify import IPAdapterAttnProcessor, IPAdapterAttnProcessor2_0
        for proc in self.unet.attn_processors.values():
            if isinstance(proc, (IPAdapterAttnProcessor, IPAdapterAttnProcessor2_0)):
                sc = scale if isinstance(scale, list) else [scale] * len(proc.scale)
                if len(proc.scale) != len(sc):
                    raise ValueError(f"`scale` should be a list of same length as the number if ip-adapters "
                                     f"Expected {len(proc.scale)} but got {len(sc)}.")
                proc.scale = sc
        self._graphs = {}                            # the scales are baked into a captured step

    def unload_ip_adapter(self):
        """ip_adapter.py:266-292: drops the image projection and restores plain processors.  (The reference's
        has-SDPA test is inverted and installs `AttnProcessor`; both classes compute the same function)"""
        from .attention_modify import AttnProcessor
        self.unet.encoder_hid_proj = None
        self.unet.config["encoder_hid_dim_type"] = None
        self.unet.set_attn_processor(AttnProcessor())
        self._graphs = {}

    def encode_image(self, image, device, num_images_per_prompt, output_hidden_states=None):
        """reference :148-170: CLIP image embeddings (or penultimate hidden states) of the IP-Adapter image and their
        unconditional counterpart; `image_encoder` is any transformers-style CLIPVisionModelWithProjection, `feature_extractor`
        its image processor"""
        if self.image_encoder is None:
            raise NotImplementedError("ip_adapter_image needs the pipeline's image_encoder (a CLIP vision model with "
                                      "projection) and feature_extractor; or pass ip_adapter_image_embeds")
        dtype = next(self.image_encoder.parameters()).dtype
        if not isinstance(image, torch.Tensor):
            image = self.feature_extractor(image, return_tensors="pt").pixel_values
        image = image.to(device=device, dtype=dtype)
        if output_hidden_states:
            pos = self.image_encoder(image, output_hidden_states=True).hidden_states[-2]
            neg = self.image_encoder(torch.zeros_like(image), output_hidden_states=True).hidden_states[-2]
            return pos.repeat_interleave(num_images_per_prompt, dim=0), neg.repeat_interleave(num_images_per_prompt, dim=0)
        emb = self.image_encoder(image).image_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        return emb, torch.zeros_like(emb)

    def prepare_ip_adapter_image_embeds(self, ip_adapter_image, ip_adapter_image_embeds, device, num_images_per_prompt,
                                        do_classifier_free_guidance):
        """Reference :173-222, pre-computed-embeddings branch (:203-221): each list entry holds [negative; positive]
        along dim 0 when CFG is on; both halves are repeated per image and re-concatenated."""
        if ip_adapter_image_embeds is None:                                                       # :176-201
            if not isinstance(ip_adapter_image, list):
                ip_adapter_image = [ip_adapter_image]
            layers = self.unet.encoder_hid_proj.image_projection_layers
            if len(ip_adapter_image) != len(layers):
                raise ValueError(f"`ip_adapter_image` must have same length as the number of IP Adapters. Got "
                                 f"{len(ip_adapter_image)} images and {len(layers)} IP Adapters.")
            from .u_net_condition_modify import ImageProjection
            out = []
            for img, layer in zip(ip_adapter_image, layers):
                hidden = not isinstance(layer, ImageProjection)            # Plus-style projections take hidden states
                pos, neg = self.encode_image(img, device, 1, hidden)
                pos = torch.stack([pos] * num_images_per_prompt, dim=0)
                neg = torch.stack([neg] * num_images_per_prompt, dim=0)
                if do_classifier_free_guidance:
                    pos = torch.cat([neg, pos]).to(device)
                out.append(pos)
            return out
        out = []
        for e in ip_adapter_image_embeds:
            ones = [1] * (e.dim() - 1)
            if do_classifier_free_guidance:
                neg, pos = e.chunk(2)
                e = torch.cat([neg.repeat(num_images_per_prompt, *ones), pos.repeat(num_images_per_prompt, *ones)])
            else:
                e = e.repeat(num_images_per_prompt, *ones)
            out.append(e)
        return out

    # ------------------------------------------------------------------ txt2img
    @torch.no_grad()
    def txt2img(self, prompt: Union[str, List[str], None] = None, height: int = 512, width: int = 512,
                num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None, eta: float = 0.0,
                generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                output_type: Optional[str] = "pil", callback_steps: Optional[int] = 1, upscale=False, upscale_x: float = 2.0,
                upscale_method: str = "bicubic", upscale_antialias: bool = False, upscale_denoising_strength: float = 0.7,
                region_map_state=None, sampler_name="", sampler_opt={}, sampler_name_hires="", sampler_opt_hires={},
                start_time=-1, timeout=180,
                latent_processing=0, weight_func=lambda w, sigma, qk: w * sigma * qk.std(), seed=0,
                ip_adapter_image=None, control_img=None, controlnet_conditioning_scale=None, control_guidance_start=None,
                control_guidance_end=None, image_t2i_adapter=None, adapter_conditioning_scale=1.0,
                adapter_conditioning_factor: float = 1.0, guidance_rescale: float = 0.0,
                cross_attention_kwargs=None, clip_skip=None, long_encode=0, num_images_per_prompt=1,
                ip_adapter_image_embeds=None,
                # build-specific inputs (the prompt encoders are a "next" row):
                prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                text_input_ids=None, fused: Optional[bool] = None, slot: int = 0, **unsupported):
        hires = dict(prompt=prompt, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                     negative_prompt=negative_prompt, generator=generator, strength=upscale_denoising_strength,
                     sampler_name=sampler_name_hires or sampler_name, sampler_opt=sampler_opt_hires or sampler_opt,
                     region_map_state=region_map_state, seed=seed, control_img=control_img,
                     latent_processing=unsupported.get("latent_upscale_processing", False),
                     controlnet_conditioning_scale=controlnet_conditioning_scale, control_guidance_start=control_guidance_start,
                     control_guidance_end=control_guidance_end, image_t2i_adapter=image_t2i_adapter,
                     adapter_conditioning_scale=adapter_conditioning_scale, adapter_conditioning_factor=adapter_conditioning_factor,
                     guidance_rescale=guidance_rescale, cross_attention_kwargs=cross_attention_kwargs, clip_skip=clip_skip,
                     long_encode=long_encode, num_images_per_prompt=num_images_per_prompt, weight_func=weight_func,
                     ip_adapter_image_embeds=ip_adapter_image_embeds, prompt_embeds=prompt_embeds,
                     negative_prompt_embeds=negative_prompt_embeds, text_input_ids=text_input_ids, output_type=output_type,
                     start_time=start_time, timeout=timeout) if upscale else None
        if image_t2i_adapter is not None:
            if getattr(self, "adapter", None) is None:
                raise ValueError("image_t2i_adapter needs an adapter: modules.t2i_adapter.setup_model_t2i_adapter(pipe, adapter)")
            from .t2i_adapter import default_height_width
            height, width = default_height_width(self, height, width, image_t2i_adapter)              # :990-991
        sampler = self.get_scheduler(sampler_name) if isinstance(sampler_name, str) else sampler_name
        device = self._execution_device
        self._do_classifier_free_guidance = guidance_scale > 1.0
        cfg = self._do_classifier_free_guidance
        if prompt_embeds is None:                                                                            # :1006-1019
            if prompt is None or self.tokenizer is None or self.text_encoder is None:
                raise NotImplementedError("pass prompt_embeds / negative_prompt_embeds / text_input_ids, or construct the "
                                          "pipeline with a tokenizer and a CLIP text encoder and pass `prompt`")
            from .encoder_prompt_modify import encode_prompt_function
            prompt_embeds, negative_prompt_embeds, text_input_ids = encode_prompt_function(
                self, prompt, device, 1, cfg, negative_prompt, clip_skip=clip_skip, long_encode=long_encode)
        n_img = prompt_embeds.shape[0] * num_images_per_prompt
        text = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        if cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds")
            text = torch.cat([negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0), text])   # :1021
        text = text.to(device=device, dtype=self.unet.dtype)
        # sigmas are cast to the model dtype on the device (:1027-1029): fp16 rounding is part of the schedule
        sigmas = self._schedule(num_inference_steps, sampler_opt, device, text.dtype)
        latents = self.prepare_latents(n_img, self.unet.config.in_channels, height, width, text.dtype, device,
                                       generator, latents)
        latents = latents * (sigmas[0] ** 2 + 1) ** 0.5                                                      # :1043
        if text_input_ids is None:
            text_input_ids = [None, None]
        region_state = encode_region_map(self, region_map_state, width=width, height=height,
                                         num_images_per_prompt=num_images_per_prompt, text_ids=text_input_ids)  # :1050
        cross_attention_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs
        added_cond_kwargs = None
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:                               # :1069-1082
            embeds = self.prepare_ip_adapter_image_embeds(ip_adapter_image, ip_adapter_image_embeds, device,
                                                          num_images_per_prompt, cfg)
            added_cond_kwargs = {"image_embeds": [e.to(device=device, dtype=text.dtype) for e in embeds]}
        self._added_cond_kwargs = added_cond_kwargs
        control_hook = self._controlnet_hook(control_img, controlnet_conditioning_scale, control_guidance_start,
                                             control_guidance_end, width, height, num_inference_steps, n_img,
                                             num_images_per_prompt, text)                                    # :1060-1061
        control_hook = self._merge_hooks(control_hook, self._adapter_hook(
            image_t2i_adapter, adapter_conditioning_scale, adapter_conditioning_factor, width, height, len(sigmas),
            num_images_per_prompt))                                                                         # :1086-1089
        preview = None
        if latent_processing == 1:                               # :1083-1084: previews of every model call's estimate
            latents_process = [self.latent_to_image(latents, output_type)]
            preview = (latents_process, output_type)
        if fused is None:
            fused = sampler is sampling.sample_dpmpp_2m and guidance_rescale == 0.0 and cfg and not self.v_prediction \
                and control_hook is None and preview is None
        if fused:
            if control_hook is not None or preview is not None:
                raise NotImplementedError("ControlNet / T2I-Adapter / latent previews run in protocol mode (fused=False)")
            latents = self._denoise_fused(latents, sigmas, text, region_state, weight_func, guidance_scale, n_img,
                                          cross_attention_kwargs, start_time, timeout, slot=slot)
        else:
            latents = self._denoise_protocol(sampler, latents, sigmas, text, region_state, weight_func, guidance_scale,
                                             guidance_rescale, n_img, cross_attention_kwargs, eta,
                                             num_inference_steps, sampler_opt, seed, start_time, timeout,
                                             control_hook=control_hook, preview=preview, slot=slot)
        if upscale:                                                                                          # :1176-1228
            res = self._hires_pass(latents, height, width, upscale_x, upscale_method, upscale_antialias, **hires)
            return latents_process + res if latent_processing == 1 else res
        if latent_processing == 1:                               # :1229-1230 (the list ends with the last model call's estimate)
            return latents_process
        return [self.latent_to_image(latents, output_type)]

    def txt2img_coalesced(self, requests, height: int = 512, width: int = 512, num_inference_steps: int = 50,
                          guidance_scale: float = 7.5, sampler_opt=None, output_type: Optional[str] = "latent",
                          weight_func=lambda w, sigma, qk: w * sigma * qk.std(), cross_attention_kwargs=None,
                          start_time=-1, timeout=180, slot: int = 0):
        """k CONCURRENT batch-1 requests - each with its own prompt rows, region masks and start latent - denoised together by ONE
        captured UNet step per sigma (serving mode; the reference serialises requests: Gradio `queue()`, app.py:3063, one latent
        per `txt2img` call).  What the requests must share is what the step itself shares: image size, step count / schedule
        (DPM++ 2M, `sampler_opt`), guidance scale.

        Every request keeps the arithmetic of its own one-image call: the rows are laid out `[u_0..u_{k-1}, c_0..c_{k-1}]`
        (model_k_diffusion.py:1021,1097), request i's std group is rows {i, k + i} (`n_std_groups = k`: the std of
        attention_modify.py:96 taken over ONE image's rows), its region tables ride on exactly those two rows
        (quirk q1: both carry the cond table), and nothing else in the UNet couples rows - so image i equals the result of
        `txt2img(**requests[i])` to launch-geometry rounding (tests/test_full_size_parity_gpu.py::
        test_coalesced_requests_equal_their_single_runs), while every kernel of the step runs on k times the rows.

        requests: list of dicts with `prompt_embeds` [1, S, ctx], `negative_prompt_embeds` [1, S, ctx], `text_input_ids`
        ([negative ids, positive ids] or None), `region_map_state` (the UI's {phrase: {map, weight, mask_outsides}} or None) and
        `latents` [1, 4, h/8, w/8] (or `generator`).  Returns the list of per-request outputs (`latent_to_image`)."""
        k = len(requests)
        if k < 1:
            raise ValueError("txt2img_coalesced: no requests")
        if guidance_scale <= 1.0 or self.v_prediction:
            raise NotImplementedError("coalesced requests run the fused classifier-free-guidance loop (guidance_scale > 1, eps-prediction)")
        device = self._execution_device
        self._do_classifier_free_guidance = True
        self._added_cond_kwargs = None
        dt = self.unet.dtype
        neg = torch.cat([r["negative_prompt_embeds"] for r in requests]).to(device=device, dtype=dt)
        pos = torch.cat([r["prompt_embeds"] for r in requests]).to(device=device, dtype=dt)
        if neg.shape != pos.shape or pos.shape[0] != k:
            raise ValueError("txt2img_coalesced: one [1, S, ctx] prompt / negative prompt pair per request, equal S")
        text = torch.cat([neg, pos])                                                                         # [u_0.., c_0..]
        sigmas = self._schedule(num_inference_steps, sampler_opt or {}, device, dt)
        lats = []
        for r in requests:
            lats.append(self.prepare_latents(1, self.unet.config.in_channels, height, width, dt, device, r.get("generator"),
                                             r.get("latents")))
        latents = torch.cat(lats) * (sigmas[0] ** 2 + 1) ** 0.5                                              # :1043
        # per-request tables {L: [2, L, S]} (rows u, c) -> {L: [2k, L, S]} in the batch's row order
        per = [encode_region_map(self, r.get("region_map_state"), width=width, height=height, num_images_per_prompt=1,
                                 text_ids=r.get("text_input_ids") or [None, None]) for r in requests]
        region_state = self._coalesce_region_tables(per)
        latents = self._denoise_fused(latents, sigmas, text, region_state, weight_func, guidance_scale, k,
                                      {} if cross_attention_kwargs is None else cross_attention_kwargs, start_time, timeout, slot=slot)
        return [self.latent_to_image(latents[i:i + 1], output_type) for i in range(k)]

    @staticmethod
    def _coalesce_region_tables(per):
        """per-request tables [{L: [2, L, S]} (rows uncond, cond) or a non-dict] -> ONE {L: [2k, L, S]} in the batch's row order
        [u_0..u_{k-1}, c_0..c_{k-1}]: the kernels read table row b for batch row b (repeat_interleave(w, H) of
        attention_modify.py:97-99 with Bw == Bc).  A request without masks rides along with zero tables (quirk q2: the region
        path with a zero bias is the plain one); no request with masks -> the first request's non-dict state (region path off)."""
        has = [isinstance(t, dict) and bool(t) for t in per]
        if not any(has):
            return per[0]
        ref = per[has.index(True)]
        per = [t if h else {L: torch.zeros_like(w) for L, w in ref.items()} for t, h in zip(per, has)]
        for t in per:
            if sorted(t) != sorted(ref) or any(t[L].shape != ref[L].shape or t[L].shape[0] != 2 for L in ref):
                raise ValueError("txt2img_coalesced: the requests' region tables differ in levels / shape (same image size and prompt length needed)")
        return {L: torch.cat([t[L][0:1] for t in per] + [t[L][1:2] for t in per]) for L in ref}

    def get_sampler_extra_args_i2i(self, sigmas, steps, sampler_opt, latents, seed, func):
        """reference :916-941"""
        params = inspect.signature(func).parameters
        extra = {}
        if "sigma_min" in params:
            extra["sigma_min"] = sigmas[-2]              # the last sigma is zero, which DPM fast / adaptive do not allow
        if "sigma_max" in params:
            extra["sigma_max"] = sigmas[0]
        if "n" in params:
            extra["n"] = len(sigmas) - 1
        if "sigma_sched" in params:
            extra["sigma_sched"] = sigmas
        if "sigmas" in params:
            extra["sigmas"] = sigmas
        if sampler_opt.get("brownian_noise", False):
            extra["noise_sampler"] = self.create_noise_sampler(latents, sigmas, steps, seed)
        if sampler_opt.get("solver_type", None) == "heun":
            extra["solver_type"] = "heun"
        return extra

    # ---- image-side helpers of img2img / inpainting
    def preprocess(self, image):
        """reference :458-481 (`[-1, 1]` NCHW float tensor; PIL images are resized down to a multiple of 8, lanczos)"""
        if isinstance(image, torch.Tensor):
            return image
        import numpy as np
        import PIL.Image
        if isinstance(image, PIL.Image.Image):
            image = [image]
        if isinstance(image[0], PIL.Image.Image):
            w, h = image[0].size
            w, h = (v - v % 8 for v in (w, h))
            arr = np.concatenate([np.array(i.convert("RGB").resize((w, h), resample=PIL.Image.LANCZOS))[None] for i in image])
            return torch.from_numpy(2.0 * (arr.astype(np.float32) / 255.0).transpose(0, 3, 1, 2) - 1.0)
        return torch.cat(list(image), dim=0)

    def _image_tensor(self, image, height, width, mask=False):
        """What the reference's VaeImageProcessor.preprocess calls (:1447-1449, :1480-1482) amount to for tensors, numpy
        arrays and PIL images: NCHW float32 at (height, width); images in [-1, 1], masks one channel binarised at 0.5"""
        import numpy as np
        if not isinstance(image, torch.Tensor):
            import PIL.Image
            if isinstance(image, PIL.Image.Image):
                image = image.convert("L" if mask else "RGB").resize((width, height), resample=PIL.Image.LANCZOS)
                image = np.array(image).astype(np.float32) / 255.0
            arr = np.asarray(image, dtype=np.float32)             # numpy: HW, HWC or NCHW, values in [0, 1]
            image = torch.from_numpy(arr)
            if arr.ndim == 2:
                image = image[None, None]
            elif arr.ndim == 3:
                image = image.permute(2, 0, 1)[None]
            if not mask:
                image = 2.0 * image - 1.0
        image = image.float()
        if image.ndim == 3:
            image = image[None]
        if mask:
            if image.shape[1] != 1:
                image = image.mean(dim=1, keepdim=True)
            image = (image >= 0.5).float()
        if tuple(image.shape[-2:]) != (height, width):
            image = torch.nn.functional.interpolate(image, size=(height, width), mode="nearest" if mask else "bilinear")
        return image

    def _encode_vae_image(self, image, generator):
        """reference :1234-1246: sample the posterior, times the scaling factor"""
        if self.vae is None or not hasattr(self.vae, "encode"):
            raise NotImplementedError("img2img / inpainting from pixels needs a VAE with an encoder "
                                      "(modules.vae_decoder.AutoencoderKL); pass 4-channel latents instead")
        image = image.to(self.vae.device, dtype=self.vae.dtype)
        return self.vae.config.scaling_factor * self.vae.encode(image).latent_dist.sample(generator)

    def _encode_text_rows(self, prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, text_input_ids,
                          num_images_per_prompt, clip_skip, long_encode, device):
        """(text rows [u.., c..] in the UNet dtype, token ids, number of images) - reference :1006-1023"""
        cfg = self.do_classifier_free_guidance
        if prompt_embeds is None:
            if prompt is None or self.tokenizer is None or self.text_encoder is None:
                raise NotImplementedError("pass prompt_embeds / negative_prompt_embeds / text_input_ids, or construct the "
                                          "pipeline with a tokenizer and a CLIP text encoder and pass `prompt`")
            from .encoder_prompt_modify import encode_prompt_function
            prompt_embeds, negative_prompt_embeds, text_input_ids = encode_prompt_function(
                self, prompt, device, 1, cfg, negative_prompt, clip_skip=clip_skip, long_encode=long_encode)
        n_img = prompt_embeds.shape[0] * num_images_per_prompt
        text = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        if cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds")
            text = torch.cat([negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0), text])
        if text_input_ids is None:
            text_input_ids = [None, None]
        return text.to(device=device, dtype=self.unet.dtype), text_input_ids, n_img

    @staticmethod
    def _randn_like_ref(shape, generator, device, dtype):
        """diffusers randn_tensor: drawn on the generator's device, moved to `device`"""
        gdev = generator.device if generator is not None else device
        return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)

    def _hires_pass(self, latents, height, width, upscale_x, upscale_method, upscale_antialias, **img2img_kwargs):
        """reference :1176-1228 (txt2img), :789-838 (img2img), :1705-1757 (inpaiting): interpolate the final latents by
        `upscale_x` and run img2img on them at `upscale_denoising_strength`"""
        f = self.vae_scale_factor
        target_height = int(height * upscale_x // f) * 8
        target_width = int(width * upscale_x // f) * 8
        latents = torch.nn.functional.interpolate(latents.float(), size=(int(target_height // f), int(target_width // f)),
                                                  mode=upscale_method,
                                                  **({"antialias": upscale_antialias} if upscale_method in ("bilinear", "bicubic") else {})
                                                  ).to(latents.dtype)
        return self.img2img(latents=latents, width=int(target_width), height=int(target_height), **img2img_kwargs)

    @torch.no_grad()
    def img2img(self, prompt=None, num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                generator: Optional[torch.Generator] = None, image=None, output_type: Optional[str] = "pil", latents=None,
                strength=1.0, region_map_state=None, sampler_name="", sampler_opt={}, start_time=-1, timeout=180,
                scale_ratio=8.0, latent_processing=0, weight_func=lambda w, sigma, qk: w * sigma * qk.std(), upscale=False,
                upscale_x: float = 2.0, upscale_method: str = "bicubic", upscale_antialias: bool = False,
                upscale_denoising_strength: float = 0.7, sampler_name_hires="", sampler_opt_hires={},
                width=None, height=None, seed=0, ip_adapter_image=None, control_img=None,
                controlnet_conditioning_scale=None, control_guidance_start=None, control_guidance_end=None,
                image_t2i_adapter=None, adapter_conditioning_scale=1.0, adapter_conditioning_factor: float = 1.0,
                guidance_rescale: float = 0.0, cross_attention_kwargs=None, clip_skip=None, long_encode=0,
                num_images_per_prompt=1, ip_adapter_image_embeds=None,
                prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                text_input_ids=None, fused: Optional[bool] = None, **unsupported):
        """reference :543-846: encode the image (or take `latents`), keep the last `strength` fraction of the schedule, add
        noise, denoise.  Reproduces the reference's start: `latents + noise * sqrt(sigma_0^2 + 1)` (:647 - sic, not
        `noise * sigma_0`)."""
        hires = dict(prompt=prompt, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                     negative_prompt=negative_prompt, generator=generator, strength=upscale_denoising_strength,
                     sampler_name=sampler_name_hires or sampler_name, sampler_opt=sampler_opt_hires or sampler_opt,
                     region_map_state=region_map_state, seed=seed, control_img=control_img,
                     latent_processing=unsupported.get("latent_upscale_processing", False),
                     controlnet_conditioning_scale=controlnet_conditioning_scale, control_guidance_start=control_guidance_start,
                     control_guidance_end=control_guidance_end, image_t2i_adapter=image_t2i_adapter,
                     adapter_conditioning_scale=adapter_conditioning_scale, adapter_conditioning_factor=adapter_conditioning_factor,
                     guidance_rescale=guidance_rescale, cross_attention_kwargs=cross_attention_kwargs, clip_skip=clip_skip,
                     long_encode=long_encode, num_images_per_prompt=num_images_per_prompt, weight_func=weight_func,
                     ip_adapter_image_embeds=ip_adapter_image_embeds, prompt_embeds=prompt_embeds,
                     negative_prompt_embeds=negative_prompt_embeds, text_input_ids=text_input_ids, output_type=output_type,
                     start_time=start_time, timeout=timeout) if upscale else None
        sampler = self.get_scheduler(sampler_name) if isinstance(sampler_name, str) else sampler_name
        device = self._execution_device
        if image is not None:
            image = self.preprocess(image)
            latents = self._encode_vae_image(image, generator)                                       # :600-606
        if latents is None:
            raise ValueError("img2img needs `image` or `latents`")
        if height is None:
            height = int(latents.shape[-2]) * 8
        if width is None:
            width = int(latents.shape[-1]) * 8
        latents = latents.to(device, dtype=self.unet.dtype)
        self._do_classifier_free_guidance = guidance_scale > 1.0
        text, text_input_ids, n_img = self._encode_text_rows(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds,
                                                             text_input_ids, num_images_per_prompt, clip_skip, long_encode,
                                                             device)
        if latents.shape[0] != n_img:
            latents = latents.repeat(n_img // latents.shape[0], 1, 1, 1)
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)                # :637-638
        t_start = max(num_inference_steps - init_timestep, 0)
        sigmas = self._schedule(num_inference_steps, sampler_opt, device, text.dtype)
        sigma_sched = sigmas[t_start:]
        noise = self._randn_like_ref(latents.shape, generator, device, text.dtype)
        latents = latents + noise * (sigma_sched[0] ** 2 + 1) ** 0.5                                 # :647
        region_state = encode_region_map(self, region_map_state, width=width, height=height,
                                         num_images_per_prompt=num_images_per_prompt, text_ids=text_input_ids)
        cross_attention_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs
        self._added_cond_kwargs = None
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            embeds = self.prepare_ip_adapter_image_embeds(ip_adapter_image, ip_adapter_image_embeds, device,
                                                          num_images_per_prompt, self.do_classifier_free_guidance)
            self._added_cond_kwargs = {"image_embeds": [e.to(device=device, dtype=text.dtype) for e in embeds]}
        control_hook = self._controlnet_hook(control_img, controlnet_conditioning_scale, control_guidance_start,
                                             control_guidance_end, width, height, len(sigma_sched), n_img,
                                             num_images_per_prompt, text)                                    # :675-676
        control_hook = self._merge_hooks(control_hook, self._adapter_hook(
            image_t2i_adapter, adapter_conditioning_scale, adapter_conditioning_factor, width, height, len(sigma_sched),
            num_images_per_prompt))                                                                         # :700-703
        preview = None
        if latent_processing == 1:                               # :696-697
            latents_process = [self.latent_to_image(latents, output_type)]
            preview = (latents_process, output_type)
        if fused is None:
            fused = sampler is sampling.sample_dpmpp_2m and guidance_rescale == 0.0 and self.do_classifier_free_guidance \
                and not self.v_prediction and control_hook is None and preview is None
        if fused:
            if control_hook is not None or preview is not None:
                raise NotImplementedError("ControlNet / T2I-Adapter / latent previews run in protocol mode (fused=False)")
            latents = self._denoise_fused(latents, sigma_sched, text, region_state, weight_func, guidance_scale, n_img,
                                          cross_attention_kwargs, start_time, timeout)
        else:
            args = self.get_sampler_extra_args_i2i(sigma_sched, len(sigma_sched), sampler_opt, latents, seed, sampler)
            latents = self._denoise_protocol(sampler, latents, sigma_sched, text, region_state, weight_func, guidance_scale,
                                             guidance_rescale, n_img, cross_attention_kwargs, 0.0, len(sigma_sched),
                                             sampler_opt, seed, start_time, timeout, sampler_args=args,
                                             control_hook=control_hook, preview=preview)
        if upscale:                                                                                          # :789-838
            res = self._hires_pass(latents, height, width, upscale_x, upscale_method, upscale_antialias, **hires)
            return latents_process + res if latent_processing == 1 else res
        if latent_processing == 1:                               # :841-842
            return latents_process
        return [self.latent_to_image(latents, output_type)]

    def _sigma_to_alpha_sigma_t(self, sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)                                                      # :1293-1297
        return alpha_t, sigma * alpha_t

    def add_noise(self, init_latents_proper, noise, sigma):
        if isinstance(sigma, torch.Tensor) and sigma.numel() > 1:                                    # :1299-1304
            sigma = sigma.sort(descending=True)[0][0].item()
        return init_latents_proper + sigma * noise

    def prepare_latents_inpating(self, batch_size, num_channels_latents, height, width, dtype, device, generator,
                                 latents=None, image=None, sigma=None, is_strength_max=True, return_noise=False,
                                 return_image_latents=False):
        """reference :1306-1362 (same spelling)"""
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if (image is None or sigma is None) and not is_strength_max:
            raise ValueError("Since strength < 1. initial latents are to be initialised as a combination of Image + Noise."
                             "However, either the image or the noise sigma has not been provided.")
        image_latents = None
        if return_image_latents or (latents is None and not is_strength_max):
            image = image.to(device=device, dtype=dtype)
            image_latents = image if image.shape[1] == 4 else self._encode_vae_image(image, generator).to(device, dtype)
            image_latents = image_latents.repeat(batch_size // image_latents.shape[0], 1, 1, 1)
        if latents is None:
            noise = self._randn_like_ref(shape, generator, device, dtype)
            latents = noise if is_strength_max else self.add_noise(image_latents, noise, sigma)
            latents = latents * (sigma.item() ** 2 + 1) ** 0.5 if is_strength_max else latents
        else:
            noise = latents.to(device)
            latents = noise * (sigma.item() ** 2 + 1) ** 0.5
        out = (latents,)
        if return_noise:
            out += (noise,)
        if return_image_latents:
            out += (image_latents,)
        return out

    @torch.no_grad()
    def inpaiting(self, prompt=None, height: int = 512, width: int = 512, num_inference_steps: int = 50,
                  guidance_scale: float = 7.5, negative_prompt=None, eta: float = 0.0,
                  generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                  output_type: Optional[str] = "pil", callback_steps: Optional[int] = 1, upscale=False,
                  region_map_state=None, sampler_name="", sampler_opt={}, start_time=-1, timeout=180,
                  latent_processing=0, weight_func=lambda w, sigma, qk: w * sigma * qk.std(), seed=0,
                  ip_adapter_image=None, control_img=None, controlnet_conditioning_scale=None, control_guidance_start=None,
                  control_guidance_end=None, image_t2i_adapter=None, adapter_conditioning_scale=1.0,
                  adapter_conditioning_factor: float = 1.0, image=None, mask_image=None,
                  masked_image_latents=None, padding_mask_crop=None, strength: float = 1.0, guidance_rescale: float = 0.0,
                  cross_attention_kwargs=None, clip_skip=None, long_encode=0, num_images_per_prompt=1,
                  ip_adapter_image_embeds=None,
                  prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                  text_input_ids=None, **unsupported):
        """reference :1365-1760 (method name as spelled there).  4-channel UNet: the known region `image_latents + sigma * noise` is
        re-imposed on the model input before every model call after the first (:1599-1612).  9-channel UNet (the inpainting
        checkpoints): mask and masked-image latents are concatenated to the duplicated latent (:1617-1619) - eager protocol mode."""
        if upscale or padding_mask_crop is not None:
            raise NotImplementedError("hires upscale after inpainting and mask cropping are outside the denoising hot path "
                                      "built here")
        num_channels_unet = self.unet.config.in_channels
        if num_channels_unet not in (4, 9):
            raise ValueError(f"The unet {self.unet.__class__} should have either 4 or 9 input channels, not {num_channels_unet}.")
        if image is None or mask_image is None:
            raise ValueError("inpaiting needs `image` and `mask_image`")
        sampler = self.get_scheduler(sampler_name) if isinstance(sampler_name, str) else sampler_name
        device = self._execution_device
        self._do_classifier_free_guidance = guidance_scale > 1.0
        text, text_input_ids, n_img = self._encode_text_rows(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds,
                                                             text_input_ids, num_images_per_prompt, clip_skip, long_encode,
                                                             device)
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)                # :1432-1438
        t_start = max(num_inference_steps - init_timestep, 0)
        sigmas = self._schedule(num_inference_steps, sampler_opt, device, text.dtype)
        sigmas = sigmas[t_start:] if 0 <= strength < 1.0 else sigmas
        is_strength_max = strength == 1.0
        init_image = self._image_tensor(image, height, width) if not (isinstance(image, torch.Tensor) and image.shape[1] == 4) \
            else image.float()
        latents, noise_inp, image_latents = self.prepare_latents_inpating(
            n_img, 4, height, width, text.dtype, device, generator, latents, image=init_image, sigma=sigmas[0],
            is_strength_max=is_strength_max, return_noise=True, return_image_latents=True)           # :1459-1478
        mask = self._image_tensor(mask_image, height, width, mask=True)
        extra_input = None
        if num_channels_unet == 9:                               # :1253-1290, :1617-1619: [latents | mask | masked-image latents]
            if masked_image_latents is None:
                if init_image.shape[1] == 4:
                    raise ValueError("the 9-channel UNet needs pixels (or masked_image_latents) to build the masked image")
                masked = init_image * (mask < 0.5)
                masked_image_latents = self._encode_vae_image(masked, generator)
            mil = masked_image_latents.to(device=device, dtype=text.dtype)
            if mil.shape[0] < n_img:
                mil = mil.repeat(n_img // mil.shape[0], 1, 1, 1)
        mask = torch.nn.functional.interpolate(mask, size=(height // self.vae_scale_factor, width // self.vae_scale_factor))
        mask = mask.to(device=device, dtype=text.dtype)                                              # :1253-1257
        if mask.shape[0] < n_img:
            mask = mask.repeat(n_img // mask.shape[0], 1, 1, 1)
        if num_channels_unet == 9:
            rows = 2 if self.do_classifier_free_guidance else 1
            extra_input = torch.cat([torch.cat([mask] * rows), torch.cat([mil] * rows)], dim=1)
        region_state = encode_region_map(self, region_map_state, width=width, height=height,
                                         num_images_per_prompt=num_images_per_prompt, text_ids=text_input_ids)
        cross_attention_kwargs = {} if cross_attention_kwargs is None else cross_attention_kwargs
        self._added_cond_kwargs = None
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            embeds = self.prepare_ip_adapter_image_embeds(ip_adapter_image, ip_adapter_image_embeds, device,
                                                          num_images_per_prompt, self.do_classifier_free_guidance)
            self._added_cond_kwargs = {"image_embeds": [e.to(device=device, dtype=text.dtype) for e in embeds]}
        sig_last = float(sigmas[-1])

        def keep_known_region(x, sigma, call_index):                                                 # :1599-1612
            if call_index == 0 or num_channels_unet != 4:
                return x
            s = float(sigma[0])
            known = image_latents
            if s > sig_last:
                alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(s)
                known = alpha_t * image_latents + sigma_t * noise_inp
            rate = (s ** 2 + 1) ** 0.5
            return ((1 - mask) * known + mask * x / rate) * rate

        preview = None
        if latent_processing == 1:                               # :1584-1585
            latents_process = [self.latent_to_image(latents, output_type)]
            preview = (latents_process, output_type)
        control_hook = self._controlnet_hook(control_img, controlnet_conditioning_scale, control_guidance_start,
                                             control_guidance_end, width, height, num_inference_steps, n_img,
                                             num_images_per_prompt, text)                                    # :1577-1578
        control_hook = self._merge_hooks(control_hook, self._adapter_hook(
            image_t2i_adapter, adapter_conditioning_scale, adapter_conditioning_factor, width, height, len(sigmas),
            num_images_per_prompt))                                                                         # :1588-1591
        latents = self._denoise_protocol(sampler, latents, sigmas, text, region_state, weight_func, guidance_scale,
                                         guidance_rescale, n_img, cross_attention_kwargs, eta, num_inference_steps,
                                         sampler_opt, seed, start_time, timeout, input_hook=keep_known_region,
                                         control_hook=control_hook, extra_input=extra_input, preview=preview)
        if latent_processing == 1:                               # :1759-1760
            return latents_process
        return [self.latent_to_image(latents, output_type)]

    # ---- protocol mode: the reference's model_fn closure (:1091-1171) + sampler call (:1172-1175)
    def _denoise_protocol(self, sampler, latents, sigmas, text, region_state, weight_func, guidance_scale,
                          guidance_rescale, n_img, cross_attention_kwargs, eta, steps, sampler_opt, seed, start_time,
                          timeout, sampler_args=None, input_hook=None, control_hook=None, extra_input=None, preview=None,
                          slot=0):
        """sampler_args: the keyword arguments for `sampler` when the caller built them itself (img2img's
        get_sampler_extra_args_i2i); input_hook(x, sigma, call_index) -> x: applied to the model input (inpainting's
        re-imposition of the known region, reference :1599-1612); extra_input [rows, c, h, w]: channels concatenated to the
        duplicated latent before the denoiser (the 9-channel inpainting UNet's mask + masked-image latents, :1617-1619 - the
        denoiser's c_in then scales them too, as in the reference); preview = (list, output_type): every model call appends the
        image of its denoised estimate (`latent_processing == 1`, :1169-1170)"""
        cfg = self.do_classifier_free_guidance
        kdm = self.k_diffusion_model
        calls = [0]
        # Graph-backed model calls: when nothing has to enter the UNet per call besides (x, sigma) - no ControlNet / adapter
        # residuals, eps-prediction, CFG without rescale, fp16 - the sampler's model call is: one launch that writes the
        # static inputs, ONE replay of the captured UNet step (the fused mode's graph), one launch for CFG + eps -> denoised.
        # Every sampler then runs at graph speed; only its own update arithmetic stays eager.
        control = None if control_hook is None else getattr(control_hook, "static", None)
        use_graph = (ops.PROTOCOL_GRAPH and (control_hook is None or control is not None) and cfg and guidance_rescale == 0.0
                     and not self.v_prediction and latents.is_cuda and text.dtype == torch.float16 and extra_input is None
                     and self._added_cond_kwargs is None and latents.numel() // latents.shape[0] % 8 == 0)
        if use_graph:
            levels = tuple(sorted((int(L), tuple(w.shape)) for L, w in region_state.items())) \
                if isinstance(region_state, dict) else None
            ckey = None if control is None else tuple(
                ("cn", id(self.controlnet), tuple(tuple(i.shape) for i in (p["image"] if isinstance(p["image"], list) else [p["image"]])))
                if p["kind"] == "controlnet" else ("ad", tuple(tuple(v.shape) for v in p["state"])) for p in control)
            key = (("slot", slot), n_img, tuple(latents.shape), levels, tuple(text.shape), text.dtype,
                   self._weight_func_key(weight_func), None) + (() if ckey is None else (ckey,))
            _lib.check(_lib.load_library().dsc_set_workspace_slot(slot), "dsc_set_workspace_slot")    # see _denoise_fused
            st = self._static_step(key, n_img, tuple(latents.shape), text, region_state, weight_func, cross_attention_kwargs,
                                   control=control)
            scratch = torch.zeros_like(latents, dtype=text.dtype)

            def set_control(s):
                for part in control or []:
                    v = part["schedule"](s)
                    if part["kind"] == "controlnet":
                        for buf, val in zip(st["cn"]["scale"], v if isinstance(v, list) else [v]):
                            buf.fill_(float(val))
                    else:
                        st["ad"]["gate"].fill_(1.0 if v else 0.0)

            def model_fn(x, sigma):
                if start_time > 0 and timeout > 0:
                    assert (time.time() - start_time) < timeout, "inference process timed out"
                if input_hook is not None:
                    x = input_hook(x, sigma, calls[0])
                calls[0] += 1
                s = float(sigma[0])                              # the host needs sigma for (c_in, t): one sync per model call
                c_in, _, t = kdm.step_scalars(s)
                d = x.to(text.dtype).contiguous().clone()
                row = None
                if st["tadd"] is not None:                       # any sigma may be asked for: this call's embedding rows, now
                    row = (self.unet.temb_add_table(torch.tensor([t], dtype=torch.float32, device=d.device))[0], st["tadd"])
                ops.prepare_unet_input(d, c_in, t, s, st["x_in"], st["t"], st["sigma"], row=row)
                set_control(s)
                st["run"]()
                # a = 0, b = 1, c = 0: d <- D = x - sigma (eps_u + g (eps_c - eps_u)); the "next input" it also writes is unused
                ops.cfg_dpmpp2m_step(d, st["eps"], scratch, s, guidance_scale, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0,
                                     st["x_in"], st["t"], st["sigma"])
                if preview is not None:
                    preview[0].append(self.latent_to_image(d, preview[1]))
                return d.to(x.dtype)

            extra = sampler_args if sampler_args is not None else \
                self.get_sampler_extra_args_t2i(sigmas, eta, steps, sampler_opt, latents, seed, sampler)
            out = sampler(model_fn, latents, **extra)
            done = st.get("done")
            if done is None:
                done = st["done"] = torch.cuda.Event()
            done.record(torch.cuda.current_stream(latents.device))
            return out
        if slot:
            raise NotImplementedError("generation slots need the graph-backed model call (eps-prediction, CFG without rescale, "
                                      "fp16, no per-call UNet inputs)")

        def model_fn(x, sigma):
            if start_time > 0 and timeout > 0:
                assert (time.time() - start_time) < timeout, "inference process timed out"
            if input_hook is not None:
                x = input_hook(x, sigma, calls[0])
            calls[0] += 1
            latent_model_input = torch.cat([x] * 2) if cfg else x
            if extra_input is not None:
                latent_model_input = torch.cat([latent_model_input, extra_input.to(latent_model_input.dtype)], dim=1)
            cross_attention_kwargs["region_prompt"] = {
                "region_state": region_state, "sigma": sigma[0], "weight_func": weight_func, "n_std_groups": n_img}
            if latent_model_input.dtype != text.dtype:
                latent_model_input = latent_model_input.to(text.dtype)
            # CompVisDenoiser.forward broadcasts sigma[B] against 2B rows (external_k_diffusion.py:109-114), which
            # only works for B == 1 in the reference; repeat sigma per row so that B > 1 works too
            sig_rows = torch.cat([sigma] * 2) if cfg else sigma
            extra_kw = {} if self._added_cond_kwargs is None else {"added_cond_kwargs": self._added_cond_kwargs}
            if control_hook is not None:
                extra_kw.update(control_hook(latent_model_input, sig_rows))
            noise_pred = kdm(latent_model_input, sig_rows, cond=text, cross_attention_kwargs=cross_attention_kwargs,
                             **extra_kw)
            if cfg:
                u, c = noise_pred.chunk(2)
                noise_pred = u + guidance_scale * (c - u)
                if guidance_rescale > 0.0:
                    noise_pred = rescale_noise_cfg(noise_pred, c, guidance_rescale=guidance_rescale)
            if preview is not None:
                preview[0].append(self.latent_to_image(noise_pred, preview[1]))
            return noise_pred

        extra = sampler_args if sampler_args is not None else \
            self.get_sampler_extra_args_t2i(sigmas, eta, steps, sampler_opt, latents, seed, sampler)
        return sampler(model_fn, latents, **extra)

    # ---- fused mode
    def _control_buffers(self, control, st=None):
        """Static buffers of the ControlNet / T2I-Adapter parts of a captured step (control = the hooks' static
        descriptions): conditioning embeddings computed ONCE per generation, per-call scales / gates as 0-dim device tensors
        the model call fills before each replay.  With `st` given the existing buffers are refreshed in place."""
        from .controlnet import MultiControlNetModel
        out = {"cn": None, "ad": None} if st is None else st
        for part in control or []:
            if part["kind"] == "controlnet":
                multi = isinstance(self.controlnet, MultiControlNetModel)
                nets = list(self.controlnet.nets) if multi else [self.controlnet]
                imgs = part["image"] if multi else [part["image"]]
                embs = [n.conditioning_embedding(im) for n, im in zip(nets, imgs)]
                if st is None:
                    dev, dt = embs[0].device, embs[0].dtype
                    out["cn"] = {"emb": [e.clone() for e in embs], "img": [im.clone() for im in imgs], "multi": multi,
                                 "scale": [torch.zeros((), device=dev, dtype=dt) for _ in nets]}
                else:
                    for dst, src in zip(st["cn"]["emb"], embs):
                        dst.copy_(src)
            else:
                if st is None:
                    out["ad"] = {"state": [v.clone() for v in part["state"]],
                                 "gate": torch.zeros((), device=part["state"][0].device, dtype=part["state"][0].dtype)}
                else:
                    for dst, src in zip(st["ad"]["state"], part["state"]):
                        dst.copy_(src)
        return out

    def _static_step(self, key, *args, **kwargs):
        """_build_or_refresh_step, with first-time creation (warm-up launches + stream capture) serialised across host
        threads: a capture must not overlap another thread's allocations.  Generation slots should still be created one
        after the other before they run concurrently - a running slot's thread allocates too."""
        if key in self._graphs:
            return self._build_or_refresh_step(key, *args, **kwargs)
        with _CAPTURE_LOCK:
            return self._build_or_refresh_step(key, *args, **kwargs)

    def _build_or_refresh_step(self, key, n_img, lat_shape, text, region_state, weight_func, cross_attention_kwargs, control=None,
                               n_std_groups=None):
        """Static buffers + the captured UNet step.  The graph is keyed by SHAPES only: a new generation with other
        text / other region masks updates the static buffers in place (text, its packed K/V, the compressed region
        tables) and replays the same graph."""
        st = self._graphs.get(key)
        comp_cpu = self._compress_tables(region_state)
        ack = self._added_cond_kwargs
        # Which tables does the captured step READ?  The prepared-operand kernels read the compressed (ids, rows) buffers;
        # every other route (a table with more than 32 distinct rows, a custom weight_func, prompts beyond the chunked
        # kernels) reads the DENSE table - so the graph must be given a static device copy that each generation refreshes in
        # place, never the first generation's own tensors (a later generation with the same shapes would replay old masks).
        need_dense = isinstance(region_state, dict) and bool(region_state) and (
            comp_cpu is None or self._weight_func_key(weight_func) != "default" or text.shape[1] > 384
            or not self._all_cross_attention_packable())         # (<= 384 text keys: the chunked prepared-operand kernels)
        if (st is not None and (st["compressed"] is None) == (comp_cpu is None)
                and (st["dense"] is None) == (not need_dense) and st["profile"] == ops.tuning_profile()):
            done = st.get("done")
            if done is not None:         # the slot's buffers may last have been driven from another stream
                torch.cuda.current_stream(text.device).wait_event(done)
            if ack is not None:
                for dst, src in zip(st["image_embeds"], ack["image_embeds"]):
                    dst.copy_(src)
            st["text"].copy_(text)
            self._refresh_text_kv(st["text"])
            self._upload_tables(st, comp_cpu, region_state)
            self._control_buffers(control, st)
            return st
        dev, dt = text.device, text.dtype
        rows = 2 * n_img
        st = {
            "x_in": torch.zeros((rows,) + lat_shape[1:], device=dev, dtype=dt),
            "t": torch.zeros(rows, device=dev, dtype=torch.float32),
            "sigma": torch.ones(1, device=dev, dtype=torch.float32),
            "text": text.clone(),
            "compressed": None, "dense": None,
            "weight_func": weight_func,          # kept alive: the key holds its id
            "profile": ops.tuning_profile(),     # the launch rules baked into the capture (ops.set_tuning_profile)
            # the time-embedding projections of all ResNet blocks for the step about to run: written by the sampler kernels
            # (a row of the per-schedule table, _denoise_fused) instead of being recomputed by three launches inside every step
            "tadd": (torch.zeros((rows, self.unet.temb_width()), device=dev, dtype=dt)
                     if ops.USE_TEMB_HOIST and hasattr(self.unet, "temb_add_table") and dt == torch.float16 else None),
            "temb_tab": None, "temb_key": None,
            "image_embeds": None if ack is None else [e.clone() for e in ack["image_embeds"]],
        }
        if comp_cpu is not None:
            st["compressed"] = {L: (ids.to(dev), rws.to(dev)) for L, (ids, rws) in comp_cpu.items()}
        if need_dense:
            st["dense"] = {L: w.to(device=dev, dtype=torch.float32).contiguous().clone() for L, w in region_state.items()}
        st.update(self._control_buffers(control))
        self._refresh_text_kv(st["text"])
        kw = dict(cross_attention_kwargs)
        # without a dense copy the dict only supplies the level keys / shapes: no kernel of the captured step reads it
        # "compressed": False = the tables are static buffers refreshed in place: the processors must read them densely and
        # must not derive (and bake into the capture) a compressed form from the values they hold during the capture
        kw["region_prompt"] = {"region_state": st["dense"] if need_dense else region_state,
                               "compressed": st["compressed"] if st["compressed"] is not None else (False if need_dense else None),
                               "sigma": st["sigma"], "weight_func": weight_func,
                               "n_std_groups": n_img if n_std_groups is None else n_std_groups}

        ukw = {} if ack is None else {"added_cond_kwargs": {"image_embeds": st["image_embeds"]}}

        def step():
            extra = dict(ukw)
            if st["cn"] is not None:                # ControlNet on the same scaled input and timestep as the UNet (:1134-1142)
                cn = st["cn"]
                one = not cn["multi"]
                down, mid = self.controlnet(st["x_in"], st["t"], encoder_hidden_states=st["text"],
                                            controlnet_cond=cn["img"][0] if one else cn["img"],
                                            conditioning_scale=cn["scale"][0] if one else cn["scale"], guess_mode=False,
                                            return_dict=False, cond_embedding=cn["emb"][0] if one else cn["emb"])
                extra.update({"down_block_additional_residuals": down, "mid_block_additional_residual": mid})
            if st["ad"] is not None:                # T2I-Adapter features, gated per call (0 = the reference passes none)
                extra["down_intrablock_additional_residuals"] = [v * st["ad"]["gate"] for v in st["ad"]["state"]]
            if st["tadd"] is not None:
                extra["temb_adds"] = st["tadd"]
            if ops.USE_CFG_SHARED_PREFIX and st["cn"] is None and st["ad"] is None and hasattr(self.unet, "temb_add_table"):
                extra["cfg_shared_prefix"] = True    # x_in = [x; x] (ops.prepare_unet_input / cfg_dpmpp2m_step), one timestep
            return self.unet(st["x_in"], st["t"], encoder_hidden_states=st["text"], cross_attention_kwargs=kw, **extra).sample

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                   # warm-up: workspaces, MIOpen / hipBLASLt kernel selection
                st["eps"] = step()
        torch.cuda.current_stream(dev).wait_stream(side)
        if not ops.GRAPHS_ENABLED:
            st["run"] = lambda: st.__setitem__("eps", step())
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["eps"] = step()
            st["graph"] = g
            st["run"] = g.replay
        # keep one captured step per generation slot
        tag = self._slot_of(key)
        self._graphs = {k: v for k, v in self._graphs.items() if self._slot_of(k) != tag}
        self._graphs[key] = st
        return st

    def _all_cross_attention_packable(self):
        """every cross-attention layer's head dim has a packed text K/V image (_refresh_text_kv), i.e. runs the
        prepared-operand kernels that read the compressed table"""
        from .u_net_condition_modify import Attention
        for m in self.unet.modules():
            if isinstance(m, Attention) and m.is_cross_attention:
                d = m.to_q.weight.shape[0] // m.heads
                if d % 8 != 0 or d > 160:
                    return False
        return True

    @staticmethod
    def _weight_func_key(weight_func):
        """"default" for anything that behaves as `w * sigma * qk.std()`; otherwise by VALUE where that is safe - (code object,
        closure cell contents, defaults) when every captured value is an immutable scalar (int / float / bool / str / None or a
        tuple of those), so that app.py's fresh lambda per request (app.py:1004) re-uses the captured step instead of paying two
        warm-up steps and a capture under the capture lock per request - and by object identity otherwise: a captured MUTABLE
        object hashes by identity, two closures over it would share one captured step whose baked-in scalars go stale when the
        object changes, and the key would keep the object alive as long as the graph cache."""
        if weight_func is None or weight_func_is_default(weight_func):
            return "default"

        def immutable(v):
            return v is None or type(v) in (int, float, bool, str) or (type(v) is tuple and all(immutable(e) for e in v))

        code = getattr(weight_func, "__code__", None)
        if code is not None:
            try:
                cells = tuple(c.cell_contents for c in (getattr(weight_func, "__closure__", None) or ()))
                defaults = getattr(weight_func, "__defaults__", None) or ()
                kwdefaults = tuple(sorted((getattr(weight_func, "__kwdefaults__", None) or {}).items()))
                if all(immutable(v) for v in cells) and all(immutable(v) for v in defaults) and all(immutable(v) for _, v in kwdefaults):
                    return ("wf", code, cells, defaults, kwdefaults)
            except ValueError:                                       # empty cell
                pass
        return id(weight_func)

    @staticmethod
    def _slot_of(key):
        return key[0][1] if isinstance(key[0], tuple) and key[0][:1] == ("slot",) else 0

    @staticmethod
    def _compress_tables(region_state):
        """{L: (ids, rows padded to 32)} on the CPU, or None when there is no table / a level is not compressible
        (then the dense tables are used and the graph is re-captured per generation)."""
        if not isinstance(region_state, dict) or not region_state:
            return None
        out = {}
        for L, w in region_state.items():
            c = ops.compress_region_table(w.float().cpu() if w.is_cuda else w.float(), pad_rows=True)
            if c is None:
                return None
            if c[1].shape[1] <= 96:            # one text chunk: the rows in the forward kernel's own table shape (flat 16-byte copy)
                c = (c[0], ops.pad_region_rows(c[1]))
            out[L] = c
        return out

    @staticmethod
    def _upload_tables(st, comp_cpu, region_state):
        if st["dense"] is not None:              # the dense tables the captured step reads: refreshed in place
            for L, dst in st["dense"].items():
                src = region_state[L]
                dst.copy_(src.pin_memory() if not src.is_cuda else src, non_blocking=True)
        if comp_cpu is None:
            return
        for L, (ids, rws) in comp_cpu.items():
            dst_ids, dst_rows = st["compressed"][L]
            # pinned staging: a copy from pageable memory waits for the stream, i.e. for the previous generation
            dst_ids.copy_(ids.pin_memory() if dst_ids.is_cuda else ids, non_blocking=True)
            dst_rows.copy_(rws.pin_memory() if dst_rows.is_cuda else rws, non_blocking=True)

    def _refresh_text_kv(self, text):
        """K/V projections of the text for every cross-attention layer, once per generation (they are step-invariant;
        the reference recomputes them in all 16 layers x 25 steps).  Written IN PLACE into the buffers the captured
        graph reads."""
        from .u_net_condition_modify import Attention
        for m in self.unet.modules():
            if isinstance(m, Attention) and m.is_cross_attention:
                if text.is_cuda and text.dtype == torch.float16 and type(m.to_k) is torch.nn.Linear and m.to_k.bias is None:
                    # the package's own GEMM (as every linear of the step): the same bits in every process / on every rank
                    k = ops.linear(text, m.to_k.weight, prefer_kernel=True)
                    v = ops.linear(text, m.to_v.weight, prefer_kernel=True)
                else:
                    k, v = m.to_k(text), m.to_v(text)
                B, S, C = k.shape
                d = C // m.heads
                # one entry per static text buffer (= per generation slot); the newest few are kept
                slots = m.__dict__.setdefault("kv_caches", [])
                c = next((e for e in slots if e["src"] is text and e["k"].shape == k.shape), None)
                if c is not None:
                    c["k"].copy_(k)
                    c["v"].copy_(v)
                else:
                    c = {"src": text, "k": k, "v": v, "packed": None}
                    slots[:] = [e for e in slots if e["src"] is not text][-3:] + [c]
                m.kv_cache = c
                # MFMA-fragment image of K / V^T for the fused kernel (rewritten in place: captured graphs keep reading it)
                if S <= 384 and d % 8 == 0 and d <= 160:          # > 96 keys: one image per 96-key chunk (long prompts)
                    c["packed"] = ops.xattn_kv_pack(c["k"].view(B, S, m.heads, d), c["v"].view(B, S, m.heads, d),
                                                    out=c.get("packed"))

    def _drop_text_kv(self):
        from .u_net_condition_modify import Attention
        for m in self.unet.modules():
            if isinstance(m, Attention):
                m.kv_cache = None
                m.__dict__.pop("kv_caches", None)

    def _denoise_fused(self, latents, sigmas, text, region_state, weight_func, guidance_scale, n_img,
                       cross_attention_kwargs, start_time, timeout, slot=0):
        if self.v_prediction:
            raise NotImplementedError("the fused step (dsc_cfg_dpmpp2m_step) computes denoised = x - sigma * eps; "
                                      "v-prediction models run in protocol mode (fused=False)")
        prof = os.environ.get("DSC_PROFILE_HOST") == "1"
        if prof:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        kdm = self.k_diffusion_model
        sig = getattr(sigmas, "_dsc_host", None)                     # host copy made where the schedule was built, or
        if sig is None:
            sig = sigmas.detach().float().cpu().tolist()             # one device->host transfer before the loop
        coeffs = sampling.dpmpp_2m_coefficients(sig)
        levels = tuple(sorted((int(L), tuple(w.shape)) for L, w in region_state.items())) \
            if isinstance(region_state, dict) else None
        ack = getattr(self, "_added_cond_kwargs", None)
        ip_key = None if ack is None else (tuple(tuple(e.shape) for e in ack["image_embeds"]),
                                           id(getattr(self.unet, "encoder_hid_proj", None)))
        # a callable that behaves as the default `w * sigma * qk.std()` (app.py:1004 builds a fresh lambda per request) runs
        # the fused kernels whatever object it is; any other callable is baked into the captured step, so the key is the
        # OBJECT (two closures of one code object may capture different values) and the step keeps it alive
        key = (("slot", slot), n_img, tuple(latents.shape), levels, tuple(text.shape), text.dtype,
               self._weight_func_key(weight_func), ip_key)
        if latents.is_cuda:
            # generation slots: each keeps its own static buffers, captured step, packed K/V and library-GEMM workspace, so
            # two generations can be in flight on two streams (one host thread per slot)
            _lib.check(_lib.load_library().dsc_set_workspace_slot(slot), "dsc_set_workspace_slot")
        st = self._static_step(key, n_img, tuple(latents.shape), text, region_state, weight_func, cross_attention_kwargs)
        x = latents.contiguous().clone()
        old = torch.zeros_like(x)
        if prof:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        c_in, _, t = kdm.step_scalars(sig[0])
        tab = None
        if st["tadd"] is not None:
            # the schedule's timesteps are known here: every step's embedding rows in one table (kept while the schedule repeats)
            tkey = (tuple(sig[:len(coeffs)]), self.unet.temb_signature())     # the schedule AND the weights the table bakes in
            if st["temb_key"] != tkey:
                ts = [float(kdm.step_scalars(s_)[2]) for s_ in sig[:len(coeffs)]]
                st["temb_tab"] = self.unet.temb_add_table(torch.tensor(ts, dtype=torch.float32, device=x.device))
                st["temb_key"] = tkey
            tab = st["temb_tab"]
        ops.prepare_unet_input(x, c_in, t, sig[0], st["x_in"], st["t"], st["sigma"], row=None if tab is None else (tab[0], st["tadd"]))
        for i, (a, b, c) in enumerate(coeffs):
            if start_time > 0 and timeout > 0:
                assert (time.time() - start_time) < timeout, "inference process timed out"
            st["run"]()
            nxt = sig[i + 1]
            c_in_n, _, t_n = kdm.step_scalars(nxt) if nxt > 0 else (1.0, 0.0, 0.0)
            # x <- a*x + b*D + c*D_old with D = x - sigma*(eps_u + g*(eps_c - eps_u)); also writes next x_in/t/sigma
            ops.cfg_dpmpp2m_step(x, st["eps"], old, sig[i], guidance_scale, a, b, c, c_in_n, t_n, max(nxt, 1e-10),
                                 st["x_in"], st["t"], st["sigma"],
                                 row=(tab[i + 1], st["tadd"]) if tab is not None and i + 1 < len(coeffs) else None)
        if x.is_cuda:
            done = st.get("done")
            if done is None:
                done = st["done"] = torch.cuda.Event()
            done.record(torch.cuda.current_stream(x.device))
        if prof:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"[dsc host profile] setup (tables, text K/V, graph) {1e3 * (t1 - t0):.1f} ms, "
                  f"{len(coeffs)}-step loop {1e3 * (t2 - t1):.1f} ms", flush=True)
        return x
