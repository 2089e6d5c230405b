"""`resize`: sample an image on a regular (separable) lattice with `grid_pull`.
Caller of the hot path; same signature and anchor conventions as the
reference's `interpol/resize.py:13-119`."""
import torch

from .api import grid_pull, spline_coeff_nd
from .separable import separable_pull
from .sepgrid import SeparableGrid
from .utils import make_list

__all__ = ['resize']


def _lattice(anchor, factor, n_in, n_out, **bck):
    """1-D sampling positions (voxels of the input) for one dimension."""
    if anchor == 'c':       # centres of the corner voxels are aligned
        return torch.linspace(0, n_in - 1, n_out, **bck)
    if anchor == 'e':       # edges of the corner voxels are aligned
        scale = n_in / n_out
        return torch.arange(0., n_out, **bck) * scale + 0.5 * (scale - 1)
    if anchor == 'f':       # first voxel aligned, exact factor
        return torch.arange(0., n_out, **bck) / factor
    if anchor == 'l':       # last voxel aligned, exact factor
        return torch.arange(0., n_out, **bck) / factor + ((n_in - 1) - (n_out - 1) / factor)
    raise ValueError('Unknown anchor {}'.format(anchor))


def resize(image, factor=None, shape=None, anchor='c', interpolation=1, prefilter=True, **kwargs):
    """Resize (batch, channel, *inshape) by `factor` and/or to `shape`.
    Defaults: bound='nearest', extrapolate=True, prefilter=True (resize.py:112-115)."""
    factor = make_list(factor) if factor else []
    shape = make_list(shape) if shape else []
    anchor = make_list(anchor)
    nb_dim = max(len(factor), len(shape), len(anchor)) or (image.dim() - 2)
    anchor = [a[0].lower() for a in make_list(anchor, nb_dim)]
    bck = dict(dtype=image.dtype, device=image.device)
    inshape = image.shape[-nb_dim:]
    if factor:
        factor = make_list(factor, nb_dim)
    elif not shape:
        raise ValueError('One of `factor` or `shape` must be provided')
    if shape:
        shape = make_list(shape, nb_dim)
    else:
        shape = [int(i * f) for i, f in zip(inshape, factor)]
    if not factor:
        factor = [o / i for o, i in zip(shape, inshape)]
    lin = [_lattice(a, f, i, o, **bck) for a, f, i, o in zip(anchor, factor, inshape, shape)]
    kwargs.setdefault('bound', 'nearest')
    kwargs.setdefault('extrapolate', True)
    kwargs.setdefault('interpolation', interpolation)
    kwargs.setdefault('prefilter', prefilter)
    # The reference stacks meshgrid_ij(*lin) into a (*shape, dim) grid and calls grid_pull
    # (resize.py:116-117).  Here: no grid tensor; floating-point images are resampled by D
    # one-dimensional passes (separable.py), label maps go through grid_pull on a SeparableGrid.
    if nb_dim <= 3 and image.dim() >= nb_dim:
        if image.dtype.is_floating_point:
            if kwargs['prefilter']:
                image = spline_coeff_nd(image, interpolation=kwargs['interpolation'], bound=kwargs['bound'], dim=nb_dim)
            return separable_pull(image, lin, kwargs['interpolation'], kwargs['bound'], kwargs['extrapolate'])
        return grid_pull(image, SeparableGrid(lin), **kwargs)
    grid = torch.stack(torch.meshgrid(*lin, indexing='ij'), dim=-1)
    return grid_pull(image, grid, **kwargs)
