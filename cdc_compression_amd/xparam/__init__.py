"""Drop-in names of the reference's xparam/modules package (unet.py, denoising_diffusion.py)."""
from ..diffusion import GaussianDiffusionX as GaussianDiffusion  # noqa: F401
from ..unet import Unet  # noqa: F401
from ..compressor import ResnetCompressor  # noqa: F401
