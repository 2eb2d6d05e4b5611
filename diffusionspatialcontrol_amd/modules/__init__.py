"""Host-side mirror of the reference's `source/modules` package for the denoising hot path."""
