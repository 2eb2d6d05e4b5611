"""diffusionspatialcontrol_amd - MI355X (gfx950) native hot path of DiffusionSpatialControl.

Host side: Python mirroring the reference's `source/modules` plug-in surface (AttnProcessor protocol,
`StableDiffusionPipeline.txt2img`).  Device side: `libdsc_hip.so`, a C-ABI library of hand-written HIP
kernels (include/dsc_hip.h), loaded with ctypes.  There is no CPU fallback: using an op without the built
library raises `DscLibraryError`.
"""
from ._lib import DscLibraryError, lib_path, load_library  # noqa: F401

__version__ = "0.1.0"
