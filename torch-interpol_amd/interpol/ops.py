"""Operator seam: non-differentiable forward / backward operators.

Same functions, argument order and tensor layouts as the reference's
`interpol/pushpull.py` (grid_pull 35-66, grid_push 70-102, grid_count 106-142,
grid_grad 146-172, grid_pushgrad 176-203, grid_hess 207-233 and the four
`*_backward` compositions 237-325):

    inp  : (B, C, *spatial_in)      grid : (B, *spatial_out, D)
    bound / interpolation : lists of int codes, padded/truncated to D
    extrapolate : 0 | 1 | 2

but every operator is ONE fused HIP kernel launch (see `csrc/`) instead of
(order+1)^D passes of gather/scatter + broadcast multiplies, and the backward
operators are fused too (no (B,C,N,D) temporary).

`use_kernels(table)` swaps the kernel table; it exists for the CPU-side tests
of the host logic (shape conventions, autograd wiring) and is never used by the
product path, whose default table is the HIP library.
"""
import contextlib

import torch

from . import _hip
from .codes import pad_codes


def _dflag(displacement):
    return _hip.FLAG_DISPLACEMENT if displacement else 0


def _kw(displacement):
    """extra kernel-table keywords (tables that predate the flag are still callable without it)"""
    return {"displacement": True} if displacement else {}


class _HipKernels:
    """Default kernel table: the gfx950 library behind the C-ABI."""

    @staticmethod
    def pull(inp, grid, bound, order, extrapolate, displacement=False):
        return _hip.gather("pull", inp, grid, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def grad(inp, grid, bound, order, extrapolate, displacement=False):
        return _hip.gather("grad", inp, grid, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def hess(inp, grid, bound, order, extrapolate, displacement=False):
        return _hip.gather("hess", inp, grid, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def push(inp, grid, shape, bound, order, extrapolate, displacement=False):
        if shape is not None and _HipKernels.expanding(inp, grid, shape, False, order):
            return _hip.push_bricks(inp, grid, shape, bound, order, extrapolate, flags=_dflag(displacement))
        return _hip.scatter("push", inp, grid, shape, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def count(grid, shape, bound, order, extrapolate, displacement=False):
        return _hip.scatter("count", None, grid, shape, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def pushgrad(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return _hip.scatter("pushgrad", inp, grid, shape, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def push_shared_(out, inp, grid, bound, order, extrapolate, with_count=False):
        """out (1,C,*shape) += sum over the batch of push(inp, grid); count when inp is None;
        with_count: out has C + 1 channels, the last one accumulates the count in the same pass."""
        op = "count" if inp is None else "push"
        shape = list(out.shape[2:])
        # Round 5: the batch items that share a target share its bricks too (csrc/push_owner.hip: BrickGrid::item = 0) -- once all
        # the sources together bring an EIGHTH of a sample per target voxel (dense_when_merged; BASELINE config 4: 64 sources of 128^3
        # into 512^3 bring a full one) the owner-computes organisation serves the call like a dense one (the probe of the call sends
        # it there): no atomics.  It needs ~22 B of workspace per sample over all sources: where that is denied the call goes on to
        # the organisations below (the target-stationary bricks first) instead of straight to the tiles.
        if _HipKernels.dense_when_merged(grid, shape, order):
            r = _hip.scatter(op, inp, grid, shape, bound, order, extrapolate,
                             flags=_hip.FLAG_ACCUMULATE, out=out, shared=True, with_count=with_count, need_workspace=True)
            if r is not None:
                return r
        if _HipKernels.expanding(inp, grid, shape, with_count, order):
            return _hip.push_bricks(inp, grid, shape, bound, order, extrapolate, flags=_hip.FLAG_ACCUMULATE, out=out,
                                    shared=True, with_count=with_count)
        return _hip.scatter(op, inp, grid, shape, bound, order, extrapolate,
                            flags=_hip.FLAG_ACCUMULATE, out=out, shared=True, with_count=with_count)

    @staticmethod
    def dense_when_merged(grid, shape, order):
        """Shared target: do the samples of ALL batch items make the owner-computes bricks worthwhile (3-D, one order 2..3)?"""
        from . import backend
        if backend.rough_deformations is False or backend.want_exact_scatter():
            return False
        if grid.shape[-1] != 3 or len(shape) != 3 or order is None or len(set(order)) != 1 or order[0] not in (2, 3):
            return False
        nsamp = int(grid.shape[0])
        for n in grid.shape[1:-1]:
            nsamp *= int(n)
        nvox = 1
        for n in shape:
            nvox *= int(n)
        return 8 * nsamp >= nvox

    @staticmethod
    def expanding(inp, grid, shape, with_count, order=None):
        """Is the target-stationary scheme (interpol_push_bricks) the better one?  Yes when the
        target has many more voxels than there are samples per item: neighbouring samples then own
        disjoint target voxels and a sample-stationary tile has nothing to merge."""
        if inp is None or not _hip.bricks_applicable(inp, grid, with_count, order):
            return False
        nsamp = 1
        for n in grid.shape[1:-1]:
            nsamp *= int(n)
        nvox = 1
        for n in shape:
            nvox *= int(n)
        return nvox >= 8 * nsamp

    @staticmethod
    def push_count(inp, grid, shape, bound, order, extrapolate, displacement=False):
        if shape is not None and _HipKernels.expanding(inp, grid, shape, True, order):
            return _hip.push_bricks(inp, grid, shape, bound, order, extrapolate, flags=_dflag(displacement), with_count=True)
        return _hip.scatter("push", inp, grid, shape, bound, order, extrapolate, flags=_dflag(displacement), with_count=True)

    @staticmethod
    def pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        return _hip.pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, flags=_dflag(displacement))

    @staticmethod
    def push_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        return _hip.push_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, flags=_dflag(displacement))

    @staticmethod
    def count_backward(grad, grid, bound, order, extrapolate, displacement=False):
        return _hip.push_backward(grad, None, grid, bound, order, extrapolate, False, True, flags=_dflag(displacement))[1]

    @staticmethod
    def spline_filter_(data, bound, order, dim, src=None):
        return _hip.spline_filter_(data, bound, order, dim, src=src)

    @staticmethod
    def pull_labels(inp, grid, bound, order, extrapolate, displacement=False):
        return _hip.pull_labels(inp, grid, bound, order, extrapolate, flags=_dflag(displacement))

    @staticmethod
    def resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint, n_lattice):
        return _hip.resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint, n_lattice)


_kernels = _HipKernels


@contextlib.contextmanager
def use_kernels(table):
    """TEST HOOK: temporarily replace the kernel table (see module docstring)."""
    global _kernels
    old = _kernels
    _kernels = table
    try:
        yield
    finally:
        _kernels = old


def kernels(*tensors, dim=None):
    """The kernel table that serves these tensors: the HIP library for CUDA tensors of 1-3 spatial dims (it raises when
    libinterpol_hip.so is missing: there is no silent fallback on the GPU), the device-generic PyTorch restatement
    (interpol/torch_kernels.py) for what the library does not cover -- CPU tensors and D > 3, which the reference's nd
    path accepts (pushpull.py:49-66).  A table installed with `use_kernels` (tests) always wins."""
    if _kernels is not _HipKernels:
        return _kernels
    on_gpu = True
    for t in tensors:
        dev = getattr(t, 'device', None)
        if dev is not None and torch.device(dev).type != 'cuda':
            on_gpu = False
    if on_gpu and (dim is None or dim <= 3):
        return _HipKernels
    from .torch_kernels import TorchKernels
    return TorchKernels


def _codes(grid, bound, interpolation):
    dim = grid.shape[-1]
    return pad_codes(bound, dim), pad_codes(interpolation, dim)


def _check_push_shapes(inp, grid, trailing=0):
    dim = grid.shape[-1]
    isp = tuple(inp.shape[2:2 + dim]) if trailing else tuple(inp.shape[-dim:])
    if isp != tuple(grid.shape[1:-1]):
        # reference interpol/iso1.py:149-150, iso0.py:82-83
        raise ValueError('Input and grid should have the same spatial shape')


def grid_pull(inp, grid, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in), (B,*out,D) -> (B,C,*out).   Reference pushpull.py:35-66."""
    bound, interpolation = _codes(grid, bound, interpolation)
    return kernels(inp, grid, dim=grid.shape[-1]).pull(inp, grid, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_push(inp, grid, shape, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in), (B,*in,D) -> (B,C,*shape).   Reference pushpull.py:70-102."""
    bound, interpolation = _codes(grid, bound, interpolation)
    _check_push_shapes(inp, grid)
    shape = None if shape is None else list(shape)
    return kernels(inp, grid, dim=grid.shape[-1]).push(inp, grid, shape, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_count(grid, shape, bound, interpolation, extrapolate, displacement=False):
    """(B,*in,D) -> (B,1,*shape).   Reference pushpull.py:106-142."""
    bound, interpolation = _codes(grid, bound, interpolation)
    shape = None if shape is None else list(shape)
    return kernels(grid, dim=grid.shape[-1]).count(grid, shape, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_grad(inp, grid, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in), (B,*out,D) -> (B,C,*out,D).   Reference pushpull.py:146-172."""
    bound, interpolation = _codes(grid, bound, interpolation)
    return kernels(inp, grid, dim=grid.shape[-1]).grad(inp, grid, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_pushgrad(inp, grid, shape, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in,D), (B,*in,D) -> (B,C,*shape).   Reference pushpull.py:176-203."""
    bound, interpolation = _codes(grid, bound, interpolation)
    _check_push_shapes(inp, grid, trailing=1)
    shape = None if shape is None else list(shape)
    return kernels(inp, grid, dim=grid.shape[-1]).pushgrad(inp, grid, shape, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_hess(inp, grid, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in), (B,*out,D) -> (B,C,*out,D,D).   Reference pushpull.py:207-233."""
    bound, interpolation = _codes(grid, bound, interpolation)
    return kernels(inp, grid, dim=grid.shape[-1]).hess(inp, grid, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_pull_backward(grad, inp, grid, bound, interpolation, extrapolate,
                       need_inp=None, need_grid=None, displacement=False):
    """-> (grad_inp (B,C,*in) | None, grad_grid (B,*out,D) | None).
    Reference pushpull.py:237-258; gradients are only computed for the inputs
    that require them (pushpull.py:252-255)."""
    bound, interpolation = _codes(grid, bound, interpolation)
    need_inp = inp.requires_grad if need_inp is None else need_inp
    need_grid = grid.requires_grad if need_grid is None else need_grid
    if not (need_inp or need_grid):
        return None, None
    return kernels(grad, inp, grid, dim=grid.shape[-1]).pull_backward(grad, inp, grid, bound, interpolation, int(extrapolate), need_inp, need_grid,
                                  **_kw(displacement))


def grid_push_backward(grad, inp, grid, bound, interpolation, extrapolate,
                       need_inp=None, need_grid=None, displacement=False):
    """-> (grad_inp (B,C,*in) | None, grad_grid (B,*in,D) | None).
    Reference pushpull.py:262-282."""
    bound, interpolation = _codes(grid, bound, interpolation)
    need_inp = inp.requires_grad if need_inp is None else need_inp
    need_grid = grid.requires_grad if need_grid is None else need_grid
    if not (need_inp or need_grid):
        return None, None
    return kernels(grad, inp, grid, dim=grid.shape[-1]).push_backward(grad, inp, grid, bound, interpolation, int(extrapolate), need_inp, need_grid,
                                  **_kw(displacement))


def grid_count_backward(grad, grid, bound, interpolation, extrapolate, need_grid=None, displacement=False):
    """-> grad_grid (B,*in,D) | None.   Reference pushpull.py:286-299."""
    bound, interpolation = _codes(grid, bound, interpolation)
    need_grid = grid.requires_grad if need_grid is None else need_grid
    if not need_grid:
        return None
    return kernels(grad, grid, dim=grid.shape[-1]).count_backward(grad, grid, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_grad_backward(grad, inp, grid, bound, interpolation, extrapolate,
                       need_inp=None, need_grid=None, displacement=False):
    """-> (grad_inp (B,C,*in) | None, grad_grid (B,*out,D) | None); only reached
    by double backward.   Reference pushpull.py:303-325."""
    dim = grid.shape[-1]
    need_inp = inp.requires_grad if need_inp is None else need_inp
    need_grid = grid.requires_grad if need_grid is None else need_grid
    grad_inp = grad_grid = None
    if need_inp:
        grad_inp = grid_pushgrad(grad, grid, inp.shape[-dim:], bound, interpolation, extrapolate, displacement)
    if need_grid:
        hess = grid_hess(inp, grid, bound, interpolation, extrapolate, displacement)
        grad_grid = (hess * grad.unsqueeze(-1)).sum(dim=[1, -2])
    return grad_inp, grad_grid


def resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint=False, n_lattice=None):
    """One 1-D pass of a tensor-product resampling along `dim` (see `separable.py`):
    forward = pull along that dim at coordinates `lin`, adjoint = the matching push."""
    return kernels(src, lin).resample1d(src, lin, dim, int(order), int(bound), int(extrapolate), int(mode), bool(adjoint), n_lattice)


def labels_covered(dim, interpolation, *tensors):
    """Can `grid_pull_labels` serve this stencil (else the caller loops over the labels)?"""
    if dim > 3 or kernels(*tensors, dim=dim) is not _HipKernels:
        return False
    return _hip.labels_covered(dim, pad_codes(interpolation, dim))


def grid_pull_labels(inp, grid, bound, interpolation, extrapolate, displacement=False):
    """Integer label map (B,C,*in), float32 grid (B,*out,D) -> int32 (B,C,*out): the label whose
    interpolated indicator image is largest (> 0), smallest label on ties, else 0 -- the result of
    the reference's loop over `input.unique()` (api.py:194-205, prefilter=False) in one pass."""
    bound, interpolation = _codes(grid, bound, interpolation)
    return kernels(inp, grid, dim=grid.shape[-1]).pull_labels(inp, grid, bound, interpolation, int(extrapolate), **_kw(displacement))


def grid_push_count(inp, grid, shape, bound, interpolation, extrapolate, displacement=False):
    """(B,C,*in), (B,*in,D) -> (B,C+1,*shape): grid_push of `inp` in channels 0..C-1 and grid_count
    of the same grid in channel C, from ONE pass over the grid (pushpull.py:70-102 + 106-142)."""
    bound, interpolation = _codes(grid, bound, interpolation)
    _check_push_shapes(inp, grid)
    shape = None if shape is None else list(shape)
    return kernels(inp, grid, dim=grid.shape[-1]).push_count(inp, grid, shape, bound, interpolation, int(extrapolate), **_kw(displacement))
