"""`restrict`: adjoint of `resize`, built on `grid_push`.  Caller of the hot
path; same signature as the reference's `interpol/restrict.py:9-122`."""
import torch

from .api import grid_push, spline_coeff_nd
from .separable import separable_push
from .sepgrid import SeparableGrid
from .utils import make_list

__all__ = ['restrict']


def restrict(image, factor=None, shape=None, anchor='c', interpolation=1, reduce_sum=False, **kwargs):
    """Restrict (batch, channel, *inshape) by `factor` (> 1 : smaller image) and/or
    to `shape`; the splat is divided by the volume ratio unless `reduce_sum`."""
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
        shape = [int(i / f) for i, f in zip(inshape, factor)]
    if not factor:
        factor = [i / o for o, i in zip(shape, inshape)]

    lin, fullscale = [], 1
    for a, f, n_in, n_out in zip(anchor, factor, inshape, shape):
        if a == 'c':
            lin.append(torch.linspace(0, n_out - 1, n_in, **bck))
            fullscale *= (n_in - 1) / (n_out - 1)
        elif a == 'e':
            scale = n_out / n_in
            lin.append(torch.arange(0., n_in, **bck) * scale + 0.5 * (scale - 1))
            fullscale *= scale
        elif a == 'f':
            lin.append(torch.arange(0., n_in, **bck) / f)
            fullscale *= 1 / f
        elif a == 'l':
            lin.append(torch.arange(0., n_in, **bck) / f + ((n_out - 1) - (n_in - 1) / f))
            fullscale *= 1 / f
        else:
            raise ValueError('Unknown anchor {}'.format(a))

    kwargs.setdefault('bound', 'nearest')
    kwargs.setdefault('extrapolate', True)
    kwargs.setdefault('interpolation', interpolation)
    kwargs.setdefault('prefilter', False)
    # reference: grid_push on stack(meshgrid_ij(*lin), -1) (restrict.py:117-118); see resize.py
    if nb_dim <= 3 and image.dim() >= nb_dim and image.dtype in (torch.float32, torch.float64):
        out = separable_push(image, lin, shape, kwargs['interpolation'], kwargs['bound'], kwargs['extrapolate'])
        if kwargs['prefilter']:
            out = spline_coeff_nd(out, interpolation=kwargs['interpolation'], bound=kwargs['bound'], dim=nb_dim, inplace=True)
    else:
        if nb_dim <= 3 and image.dim() >= nb_dim:
            grid = SeparableGrid(lin)
        else:
            grid = torch.stack(torch.meshgrid(*lin, indexing='ij'), dim=-1)
        out = grid_push(image, grid, shape, **kwargs)
    if not reduce_sum:
        out /= fullscale
    return out
