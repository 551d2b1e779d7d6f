"""cdc_compression_amd -- MI355X (gfx950) native decode hot path of CDC (conditional-diffusion image
compression): the N-step DDIM loop over the denoising U-Net, as hand-written HIP kernels behind a
C-ABI (include/cdc_hip.h), with host-side mirrors of the reference's Unet / GaussianDiffusion API."""
from .unet import Unet  # noqa: F401
from .diffusion import GaussianDiffusionEps, GaussianDiffusionX  # noqa: F401
from .compressor import BigCompressor, ResnetCompressor  # noqa: F401
from . import epsilonparam, xparam  # noqa: F401,E402  (drop-in names of the reference trees)
