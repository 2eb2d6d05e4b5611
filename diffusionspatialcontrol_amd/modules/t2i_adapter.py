"""T2I-Adapter - counterpart of reference `source/modules/t2i_adapter.py` (`_preprocess_adapter_image` :56-80,
`setup_model_t2i_adapter` :83-86, `preprocessing_t2i_adapter` :90-117, `default_height_width` :120-144) and of the diffusers
0.27.2 `T2IAdapter` / `MultiAdapter` models those functions drive (un-vendored; restated from the published structure of
Mou et al. 2023, "full_adapter" type with diffusers parameter names: `adapter.unshuffle / conv_in / body.{i}.in_conv /
body.{i}.resnets.{j}.block1|block2` - PARITY UNPINNED).

The adapter runs ONCE per generation: its four feature maps (320 / 640 / 1280 / 1280 channels at 1/8 .. 1/64 of the image)
are the `down_intrablock_additional_residuals` the UNet adds inside its down blocks (`u_net_condition_modify.py:1194-1230`)
during the first `adapter_conditioning_factor` fraction of the steps - so it is plain torch, not a kernel target.
"""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class AdapterResnetBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.block1 = nn.Conv2d(channels, channels, 3, padding=1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(channels, channels, 1)

    def forward(self, x):
        return self.block2(self.act(self.block1(x))) + x


class AdapterBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_res_blocks, down=False):
        super().__init__()
        self.downsample = nn.AvgPool2d(kernel_size=2, stride=2, ceil_mode=True) if down else None
        self.in_conv = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.resnets = nn.Sequential(*[AdapterResnetBlock(out_channels) for _ in range(num_res_blocks)])

    def forward(self, x):
        if self.downsample is not None:
            x = self.downsample(x)
        if self.in_conv is not None:
            x = self.in_conv(x)
        return self.resnets(x)


class FullAdapter(nn.Module):
    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8):
        super().__init__()
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.conv_in = nn.Conv2d(in_channels * downscale_factor ** 2, channels[0], 3, padding=1)
        self.body = nn.ModuleList([AdapterBlock(channels[0], channels[0], num_res_blocks)] +
                                  [AdapterBlock(channels[i - 1], channels[i], num_res_blocks, down=True)
                                   for i in range(1, len(channels))])
        self.total_downscale_factor = downscale_factor * 2 ** (len(channels) - 1)

    def forward(self, x) -> List[torch.Tensor]:
        x = self.conv_in(self.unshuffle(x))
        feats = []
        for blk in self.body:
            x = blk(x)
            feats.append(x)
        return feats


class T2IAdapter(nn.Module):
    """diffusers `T2IAdapter(adapter_type="full_adapter")`: image [B, in_channels, H, W] -> 4 feature maps"""

    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8,
                 adapter_type="full_adapter"):
        super().__init__()
        if adapter_type != "full_adapter":
            raise NotImplementedError("only the full_adapter type (the SD1.x adapters) is built")
        self.adapter = FullAdapter(in_channels, channels, num_res_blocks, downscale_factor)
        self.config = type("Cfg", (), {"in_channels": in_channels, "channels": tuple(channels),
                                       "downscale_factor": downscale_factor})()

    @property
    def dtype(self):
        return self.adapter.conv_in.weight.dtype

    @property
    def total_downscale_factor(self):
        return self.adapter.total_downscale_factor

    @property
    def downscale_factor(self):
        return self.adapter.unshuffle.downscale_factor

    def forward(self, x):
        return self.adapter(x)


class MultiAdapter(nn.Module):
    """diffusers `MultiAdapter`: weighted sum of several adapters' features"""

    def __init__(self, adapters):
        super().__init__()
        self.adapters = nn.ModuleList(adapters)
        self.num_adapter = len(adapters)

    @property
    def dtype(self):
        return self.adapters[0].dtype

    @property
    def downscale_factor(self):
        return self.adapters[0].downscale_factor

    def forward(self, xs, adapter_weights: Optional[List[float]] = None):
        w = [1.0 / self.num_adapter] * self.num_adapter if adapter_weights is None else \
            ([adapter_weights] * self.num_adapter if isinstance(adapter_weights, float) else list(adapter_weights))
        acc = None
        for x, wi, ad in zip(xs, w, self.adapters):
            feats = ad(x)
            acc = [wi * f for f in feats] if acc is None else [a + wi * f for a, f in zip(acc, feats)]
        return acc


def _preprocess_adapter_image(image, height, width):
    """reference :56-80: [0, 1] NCHW float32 (PIL images resized to (width, height), lanczos)"""
    if isinstance(image, torch.Tensor):
        return image
    import PIL.Image
    if isinstance(image, PIL.Image.Image):
        image = [image]
    if isinstance(image[0], PIL.Image.Image):
        arrs = [np.array(i.resize((width, height), resample=PIL.Image.LANCZOS)) for i in image]
        arrs = [a[None, ..., None] if a.ndim == 2 else a[None, ...] for a in arrs]
        return torch.from_numpy(np.concatenate(arrs, axis=0).astype(np.float32).transpose(0, 3, 1, 2) / 255.0)
    if image[0].ndim == 3:
        return torch.stack(list(image), dim=0)
    if image[0].ndim == 4:
        return torch.cat(list(image), dim=0)
    raise ValueError(f"Invalid image tensor! Expecting image tensor with 3 or 4 dimension, but recive: {image[0].ndim}")


def setup_model_t2i_adapter(class_name, adapter=None):
    """reference :83-86 (`class_name` is the pipeline object, the reference's parameter name)"""
    if isinstance(adapter, (list, tuple)):
        adapter = MultiAdapter(adapter)
    class_name.adapter = adapter


def preprocessing_t2i_adapter(class_name, image, width, height, adapter_conditioning_scale, num_images_per_prompt=1):
    """reference :90-117: the adapter's features, scaled, repeated per image and duplicated for CFG"""
    ad = class_name.adapter
    if isinstance(ad, MultiAdapter):
        inputs = [_preprocess_adapter_image(one, height, width).to(device=class_name.device, dtype=ad.dtype) for one in image]
        state = ad(inputs, adapter_conditioning_scale)
    else:
        inp = _preprocess_adapter_image(image, height, width).to(device=class_name.device, dtype=ad.dtype)
        state = [v * adapter_conditioning_scale for v in ad(inp)]
    if num_images_per_prompt > 1:
        state = [v.repeat(num_images_per_prompt, 1, 1, 1) for v in state]
    if class_name.do_classifier_free_guidance:
        state = [torch.cat([v] * 2, dim=0) for v in state]
    return state


def default_height_width(class_name, height, width, image):
    """reference :120-144: missing sizes from the (first) adapter image, rounded down to the adapter's downscale factor"""
    while isinstance(image, list):
        image = image[0]
    f = class_name.adapter.downscale_factor
    if height is None:
        height = image.height if not isinstance(image, torch.Tensor) else image.shape[-2]
        height = (height // f) * f
    if width is None:
        width = image.width if not isinstance(image, torch.Tensor) else image.shape[-1]
        width = (width // f) * f
    return height, width
