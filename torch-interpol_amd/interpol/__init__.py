"""interpol -- MI355X-native drop-in for the sampling hot path of
balbasty/torch-interpol: B-spline (order 0-7) pull / push / count / grad with
seven boundary conditions, their backward passes and the spline prefilter, as
fused hand-written HIP kernels for gfx950 behind the reference's Python API.

    import interpol
    warped = interpol.grid_pull(image, grid, interpolation=3, bound='dct2', extrapolate=True)

CUDA (ROCm) tensors of 1-3 spatial dims run the HIP kernels (libinterpol_hip.so must be built: there is no silent
fallback for them); CPU tensors and D > 3 are served by the device-generic PyTorch kernel table of interpol/torch_kernels.py.
"""
from .api import (pull, push, count, grid_pull, grid_push, grid_count, grid_grad,       # noqa: F401
                  spline_coeff, spline_coeff_nd)
from .utils import identity_grid, add_identity_grid, add_identity_grid_, affine_grid     # noqa: F401
from .resize import resize                                                                # noqa: F401
from .restrict import restrict                                                            # noqa: F401
from .sepgrid import SeparableGrid, AffineGrid                                            # noqa: F401
from .separable import separable_pull, separable_push                                    # noqa: F401
from . import backend                                                                     # noqa: F401

__version__ = "0.1.0+mi355x"
