"""Public API: `grid_pull / grid_push / grid_count / grid_grad / spline_coeff /
spline_coeff_nd` with the reference's signatures, defaults and shape
conventions (reference interpol/api.py:149-445):

    input : (..., [channel], *inshape)        grid : (..., *outshape, dim)

Batch dimensions broadcast; an input with exactly `dim` dimensions has no
channel axis.  Coordinates are in voxels, component d addresses spatial dim d.

`interpolation`: 0..7 or 'nearest' | 'linear' | 'quadratic' | 'cubic' | 'fourth'
| 'fifth' | 'sixth' | 'seventh' (or a per-dimension list).
`bound`: 'zero' | 'replicate' ('nearest') | 'dct1' ('mirror') | 'dct2' ('reflect')
| 'dst1' ('antimirror') | 'dst2' ('antireflect') | 'dft' ('wrap'), their aliases,
int codes 0..6, or a per-dimension list.
`extrapolate`: False/0 (zero outside the field of view), True/1, or 2 ('hist').

Size limits of this build (the reference has none): one (batch, channel) image -- input of the
gathers, target of the scatters -- must span less than 4 GiB of byte offsets (1024^3 fp32 is just
over, 812^3 fp64 too), spatial strides must stay below 2 GiB, D <= 3, CUDA (HIP) tensors only.
Larger problems raise `ValueError: ... bad or too large extent` from the C-ABI; split them along a
spatial axis (e.g. `interpol.distributed.grid_pull_slabs` for the sample grid) before calling.
(The statement about D and CUDA tensors holds for the HIP kernels: CPU tensors and D > 3 run the
package's PyTorch restatement, `torch_kernels.py`.)

Autograd.  First-order gradients run the fused HIP kernels.  Under `create_graph=True` (double
backward: gradient penalties, Hessian-vector products) the backward is composed from the
differentiable Functions of `autograd.py` instead, as the reference composes its backward from torch
ops (pushpull.py:237-325): the gradient with respect to the grid then materialises `grid_grad`'s
(B, C, *out, D) tensor -- C x D times the memory of the grid, and a second pass.  Third-order
derivatives and beyond through the grid (a backward of that double backward) differentiate on, as
the reference's plain-torch backward does: `GridGrad.backward` under `create_graph=True` lets
autograd differentiate the PyTorch restatement of `grid_grad` (`torch_kernels.py`) -- correct at
any order, at PyTorch speed and memory (the fused kernels carry no graph).
"""
import torch

from . import backend, ops
from .autograd import GridPull, GridPush, GridCount, GridGrad, SplineCoeff, SplineCoeffND
from .codes import bound_to_code, order_to_code, pad_codes
from .sepgrid import SeparableGrid, AffineGrid, LazyGrid
from .utils import expanded_shape

__all__ = ['pull', 'push', 'count', 'grid_pull', 'grid_push', 'grid_count', 'grid_grad',
           'spline_coeff', 'spline_coeff_nd']


def _fold(grid, input=None, mode=None):
    """Broadcast and reshape user tensors to the operator layout
    (B, C, *spatial) / (B, *spatial, dim).  Same cases as the reference's
    `_preproc` (interpol/api.py:93-130)."""
    dim = grid.shape[-1]
    if input is None and isinstance(grid, LazyGrid):     # constant lattice: no batch dims, nothing to reshape
        return grid, dict(batch=[], channel=[], dim=dim)
    if input is None:
        spatial = grid.shape[-dim - 1:-1]
        batch = grid.shape[:-dim - 1]
        info = dict(batch=list(batch), channel=[1] if batch else [], dim=dim)
        return grid.reshape([-1, *spatial, dim]), info

    sep = isinstance(grid, LazyGrid)            # constant tensor-product lattice: no batch dims
    grid_spatial = grid.shape[-dim - 1:-1]
    grid_batch = () if sep else grid.shape[:-dim - 1]
    input_spatial = input.shape[-dim:]
    channel = 0 if input.dim() == dim else input.shape[-dim - 1]
    input_batch = input.shape[:-dim - 1]
    if mode == 'push':
        grid_spatial = input_spatial = expanded_shape(grid_spatial, input_spatial)

    batch = expanded_shape(grid_batch, input_batch)
    if not sep:
        grid = grid.expand([*batch, *grid_spatial, dim]).reshape([-1, *grid_spatial, dim])
    input = input.expand([*batch, channel or 1, *input_spatial]).reshape([-1, channel or 1, *input_spatial])
    out_channel = [channel] if channel else ([1] if batch else [])
    return grid, input, dict(batch=list(batch), channel=out_channel, dim=dim)


def _unfold(out, info, mode):
    """Inverse of `_fold` on the result (reference `_postproc`, api.py:133-146)."""
    dim = info['dim']
    if mode == 'grad':
        spatial, feat = out.shape[-dim - 1:-1], [out.shape[-1]]
    else:
        spatial, feat = out.shape[-dim:], []
    return out.reshape([*info['batch'], *info['channel'], *spatial, *feat])


def _filters(interpolation, dim):
    """Does the prefilter change anything?  Orders 0 and 1 are interpolating already (coeff.py:306-307): the out-of-place
    spline_coeff_nd would only copy the image (0.18 ms for 4 x 2 x 256^3) in front of a gather that does not modify it."""
    codes = [order_to_code(o) for o in (interpolation if isinstance(interpolation, (list, tuple)) else [interpolation])]
    return max(pad_codes(codes, dim)) > 1


def grid_pull(input, grid, interpolation='linear', bound='zero', extrapolate=False, prefilter=False,
              displacement=False):
    """Sample an image at the coordinates of a deformation field.

    input (..., [channel], *inshape), grid (..., *outshape, dim) -> (..., [channel], *outshape).
    Non floating-point inputs are treated as label maps: every label is
    resampled as a soft label and the arg-max is returned (api.py:194-205).

    `displacement=True` (extension, also on grid_push / grid_count / grid_grad): `grid` holds
    voxel displacements and the identity lattice is added inside the kernel -- the fused form of
    `grid_pull(input, add_identity_grid(disp))` (api.py:490-531), same values, one tensor pass less.
    """
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    grid, input, info = _fold(grid, input)
    batch, channel = input.shape[:2]
    dim = grid.shape[-1]

    if not input.dtype.is_floating_point:
        codes = [order_to_code(o) for o in (interpolation if isinstance(interpolation, (list, tuple)) else [interpolation])]
        fused = (grid.dtype == torch.float32 and ops.labels_covered(dim, codes, input, grid)
                 and (not prefilter or max(pad_codes(codes, dim)) <= 1)          # order <= 1: the prefilter is the identity
                 and (input.dtype in (torch.bool, torch.uint8, torch.int8, torch.int16, torch.int32)
                      or (input.dtype == torch.int64 and (input.numel() == 0 or int(input.abs().max()) < 2 ** 31))))
        if fused:
            # one pass: arg-max over the labels under each stencil (csrc/labels.hip)
            out = ops.grid_pull_labels(input, grid, [bound_to_code(b) for b in (bound if isinstance(bound, (list, tuple)) else [bound])],
                                       codes, int(extrapolate), displacement).to(input.dtype)
            return _unfold(out, info, 'pull')
        out = input.new_zeros([batch, channel, *grid.shape[1:-1]])
        pmax = grid.new_zeros([batch, channel, *grid.shape[1:-1]])
        for label in input.unique():
            soft = (input == label).to(grid.dtype)
            if prefilter:
                soft = spline_coeff_nd(soft, interpolation=interpolation, bound=bound, dim=dim, inplace=True)
            soft = GridPull.apply(soft, grid, interpolation, bound, extrapolate, displacement)
            out[soft > pmax] = label
            pmax = torch.max(pmax, soft)
    else:
        if prefilter and _filters(interpolation, dim):
            input = spline_coeff_nd(input, interpolation=interpolation, bound=bound, dim=dim)
        out = GridPull.apply(input, grid, interpolation, bound, extrapolate, displacement)
    return _unfold(out, info, 'pull')


def grid_push(input, grid, shape=None, interpolation='linear', bound='zero', extrapolate=False,
              prefilter=False, displacement=False):
    """Splat an image along a deformation field (adjoint of `grid_pull`).

    input (..., [channel], *inshape), grid (..., *inshape, dim) -> (..., [channel], *shape);
    `shape` defaults to `inshape`.
    """
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    grid, input, info = _fold(grid, input, mode='push')
    dim = grid.shape[-1]
    if shape is None:
        shape = tuple(input.shape[2:])
    out = GridPush.apply(input, grid, shape, interpolation, bound, extrapolate, displacement)
    if prefilter:
        out = spline_coeff_nd(out, interpolation=interpolation, bound=bound, dim=dim, inplace=True)
    return _unfold(out, info, 'push')


def grid_count(grid, shape=None, interpolation='linear', bound='zero', extrapolate=False, displacement=False):
    """Splat ones: grid (..., *inshape, dim) -> (..., [1], *shape)."""
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    grid, info = _fold(grid)
    out = GridCount.apply(grid, shape, interpolation, bound, extrapolate, displacement)
    return _unfold(out, info, 'count')


def grid_grad(input, grid, interpolation='linear', bound='zero', extrapolate=False, prefilter=False,
              displacement=False):
    """Sample the spatial gradient of an image (voxel units):
    input (..., [channel], *inshape), grid (..., *outshape, dim) -> (..., [channel], *outshape, dim)."""
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    grid, input, info = _fold(grid, input)
    dim = grid.shape[-1]
    if prefilter and _filters(interpolation, dim):
        input = spline_coeff_nd(input, interpolation, bound, dim)
    out = GridGrad.apply(input, grid, interpolation, bound, extrapolate, displacement)
    return _unfold(out, info, 'grad')


def spline_coeff(input, interpolation='linear', bound='dct2', dim=-1, inplace=False):
    """Interpolating B-spline coefficients along ONE dimension.
    Bounds: zero (-> dct1), replicate (-> dct2), dct1, dct2, dft; dst1/dst2 raise
    NotImplementedError (reference interpol/coeff.py:231-254)."""
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    return SplineCoeff.apply(input, bound, interpolation, dim, inplace)


def spline_coeff_nd(input, interpolation='linear', bound='dct2', dim=None, inplace=False):
    """Interpolating B-spline coefficients along the last `dim` dimensions
    (ALL dimensions when `dim` is None)."""
    if backend.jitfields:
        raise RuntimeError('the jitfields backend is not part of the MI355X build')
    return SplineCoeffND.apply(input, bound, interpolation, dim, inplace)


pull = grid_pull
push = grid_push
count = grid_count
