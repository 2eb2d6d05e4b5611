"""CPU oracle - TEST INFRASTRUCTURE ONLY.

A plain torch-fp32 / numpy restatement of the reference's spatially-controlled denoising path
(duongve13112002/DiffusionSpatialControl @ 2025-02-02), each function citing the reference file:line it
follows.  It is pinned against golden vectors captured by running the reference's own Python
(tests/golden/make_golden.py -> tests/golden/*.npz); what no reference file can pin (the un-vendored
k_diffusion sampler, the diffusers UNet blocks, OpenCV's bicubic kernel) is marked "parity unpinned" where
it is defined.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as
the checker or the timed CPU baseline.  Nothing under diffusionspatialcontrol_amd/ imports it: the product
path fails loudly when the HIP library is missing.
"""
