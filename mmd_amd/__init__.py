"""mmd_amd -- MI355X (gfx950) guided-diffusion trajectory sampler behind yoraish/mmd's MPD / MPDEnsemble API.

Host-side mirrors of the reference interfaces over the C ABI of libmmd_amd.so (include/mmd_amd.h).
Compute lives in hand-written HIP kernels (mmd_amd/csrc); there is no CPU fallback."""
from .unet_spec import UNET_DIM_MULTS, unet_param_spec   # noqa: F401

__all__ = ["UNET_DIM_MULTS", "unet_param_spec"]
