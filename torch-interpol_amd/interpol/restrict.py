"""`restrict`: adjoint of `resize`, built on `grid_push`.  Caller of the hot
path; same signature as the reference's `interpol/restrict.py:9-122`."""
import torch

from .api import grid_push, spline_coeff_nd
from .separable import separable_push
from .sepgrid import SeparableGrid
from . import lattice

__all__ = ['restrict']


def restrict(image, factor=None, shape=None, anchor='c', interpolation=1, reduce_sum=False, **kwargs):
    """Restrict (batch, channel, *inshape) by `factor` (> 1 : smaller image) and/or
    to `shape`; the splat is divided by the volume ratio unless `reduce_sum`."""
    nb_dim, letters, ratios, inshape, shape = lattice.plan(image, factor, shape, anchor, shrink=True)
    bck = dict(dtype=image.dtype, device=image.device)
    # input points in the output's voxel coordinates; the splat is normalised by the volume ratio the reference uses:
    # (n_in - 1) / (n_out - 1) for centre-aligned lattices, the spacing of the points otherwise (restrict.py:92-110)
    lin, fullscale = [], 1
    for a, r, n_in, n_out in zip(letters, ratios, inshape, shape):
        x, spacing = lattice.positions(a, r, n_in, n_out, **bck)
        lin.append(x)
        fullscale *= (n_in - 1) / (n_out - 1) if a == 'c' else spacing

    kwargs.setdefault('bound', 'nearest')
    kwargs.setdefault('extrapolate', True)
    kwargs.setdefault('interpolation', interpolation)
    kwargs.setdefault('prefilter', False)
    # reference: grid_push on stack(meshgrid_ij(*lin), -1) (restrict.py:117-118); see resize.py
    if nb_dim <= 3 and image.dim() >= nb_dim and image.is_cuda and image.dtype in (torch.float32, torch.float64):
        out = separable_push(image, lin, shape, kwargs['interpolation'], kwargs['bound'], kwargs['extrapolate'])
        if kwargs['prefilter']:
            out = spline_coeff_nd(out, interpolation=kwargs['interpolation'], bound=kwargs['bound'], dim=nb_dim, inplace=True)
    else:
        if nb_dim <= 3 and image.dim() >= nb_dim and image.is_cuda:
            grid = SeparableGrid(lin)
        else:
            grid = torch.stack(torch.meshgrid(*lin, indexing='ij'), dim=-1)
        out = grid_push(image, grid, shape, **kwargs)
    if not reduce_sum:
        out /= fullscale
    return out
