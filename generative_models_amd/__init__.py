"""MI355X-native DDPM training + sampling path behind the `GM` plugin surface of matwilso/generative_models.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/gmk.h), the ctypes binding,
and the host-side mirror of the reference interface (`common.GM`, `diffusion.DiffusionModel`, `main`).
"""
__all__ = ["common"]
