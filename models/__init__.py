"""Drop-in replacement of the reference's `models` package (cvg/diffmvs models/__init__.py:1-2):
same public names, constructor arguments, forward() signatures and checkpoint layout; the
arithmetic runs in the gfx950 kernels of diffmvs_amd/libdmvs_hip.so."""
from .diffusion import CasDiffMVS
from .loss import compute_inverse_loss

__all__ = ["CasDiffMVS", "compute_inverse_loss"]
