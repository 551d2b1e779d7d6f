"""Drop-in names of the reference's epsilonparam/modules package."""
from ..diffusion import GaussianDiffusionEps as GaussianDiffusion  # noqa: F401
from ..unet import Unet as _Unet


class Unet(_Unet):
    """epsilonparam/modules/unet.py:17-27: same network, no `embd_type` argument."""

    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), context_dim_mults=(1, 2, 3, 3),
                 channels=3, context_channels=3, with_time_emb=True, device=0):
        super().__init__(dim, out_dim, dim_mults, context_dim_mults, channels, context_channels,
                         with_time_emb, "01", device)
from ..compressor import BigCompressor  # noqa: F401
