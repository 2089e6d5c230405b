"""`resize`: sample an image on a regular (separable) lattice with `grid_pull`.
Caller of the hot path; same signature and anchor conventions as the
reference's `interpol/resize.py:13-119`."""
import torch

from .api import grid_pull, spline_coeff_nd
from .separable import separable_pull
from .sepgrid import SeparableGrid
from . import lattice

__all__ = ['resize']


def resize(image, factor=None, shape=None, anchor='c', interpolation=1, prefilter=True, **kwargs):
    """Resize (batch, channel, *inshape) by `factor` and/or to `shape`.
    Defaults: bound='nearest', extrapolate=True, prefilter=True (resize.py:112-115)."""
    nb_dim, letters, ratios, inshape, shape = lattice.plan(image, factor, shape, anchor, shrink=False)
    bck = dict(dtype=image.dtype, device=image.device)
    # output points in the input's voxel coordinates
    lin = [lattice.positions(a, r, n_out, n_in, **bck)[0] for a, r, n_in, n_out in zip(letters, ratios, inshape, shape)]
    kwargs.setdefault('bound', 'nearest')
    kwargs.setdefault('extrapolate', True)
    kwargs.setdefault('interpolation', interpolation)
    kwargs.setdefault('prefilter', prefilter)
    # The reference stacks meshgrid_ij(*lin) into a (*shape, dim) grid and calls grid_pull
    # (resize.py:116-117).  Here: no grid tensor; floating-point images are resampled by D
    # one-dimensional passes (separable.py), label maps go through grid_pull on a SeparableGrid.
    if nb_dim <= 3 and image.dim() >= nb_dim and image.is_cuda:
        if image.dtype.is_floating_point:
            if kwargs['prefilter']:
                image = spline_coeff_nd(image, interpolation=kwargs['interpolation'], bound=kwargs['bound'], dim=nb_dim)
            return separable_pull(image, lin, kwargs['interpolation'], kwargs['bound'], kwargs['extrapolate'])
        return grid_pull(image, SeparableGrid(lin), **kwargs)
    grid = torch.stack(torch.meshgrid(*lin, indexing='ij'), dim=-1)
    return grid_pull(image, grid, **kwargs)
