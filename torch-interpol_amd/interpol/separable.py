"""Tensor-product resampling as D one-dimensional passes.

`resize` / `restrict` sample on `stack(meshgrid_ij(*lin), -1)` (reference
interpol/resize.py:96-117, restrict.py:88-118) through the D-dimensional
grid_pull / grid_push: (K+1)^D taps per voxel.  On such a lattice the stencil
factorises -- weights, boundary indices / signs and the extrapolation mask are
per-dimension quantities (nd.py:39-77) -- so D passes of K+1 taps give the same
operator (rounding differs by a few ulp: the sums are associated per dim).

`separable_pull` / `separable_push` are differentiable w.r.t. the image (each is
the adjoint of the other).
"""
import torch

from . import ops
from .codes import bound_to_code, order_to_code, pad_codes
from .sepgrid import SeparableGrid

__all__ = ['separable_pull', 'separable_push']


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _mode(orders):
    """iso1 / iso0 / nd semantics are decided by ALL dims (pushpull.py:48-66)."""
    if all(o == 1 for o in orders):
        return 1
    if all(o == 0 for o in orders):
        return 2
    return 0


class _SepPull(torch.autograd.Function):
    """D forward passes (interpol_resample_1d); backward = `_SepPush` on the same lattice."""

    @staticmethod
    def forward(ctx, x, lin, orders, bounds, extrapolate):
        D, mode = len(lin), _mode(orders)
        ctx.args = (lin, orders, bounds, extrapolate, tuple(x.shape[-D:]))
        out = x
        for d in reversed(range(D)):      # fastest dim first: the tensor is smallest while lanes run along samples
            out = ops.resample1d(out, lin[d], d - D, orders[d], bounds[d], extrapolate, mode)
        return out

    @staticmethod
    def backward(ctx, grad):
        lin, orders, bounds, extrapolate, inshape = ctx.args
        gx = _SepPush.apply(grad, lin, inshape, orders, bounds, extrapolate) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, None


def _gathers(x, lin):
    """Do the adjoint passes of `x` along its last len(lin) dims all take the gathering kernel (csrc/resample1d.hip:
    resample1d_adj_gather)?  The rule is the library's (interpol_resample_1d_gathers, include/interpol_hip.h): asked, not repeated.
    (A `lin` that is not non-decreasing is still served correctly by that kernel -- it detects it on the device and visits every
    sample -- but slowly; resize / restrict, the callers, build increasing lattices.)"""
    if not x.is_cuda:
        return False
    from . import _hip
    code = _hip._DTYPE_CODE.get(x.dtype)
    if code is None:
        return False
    D = len(lin)
    for d in range(D):
        inner = 1
        for e in range(d + 1, D):
            inner *= x.shape[e - D]
        if not _hip.lib().interpol_resample_1d_gathers(code, lin[d].numel(), inner):
            return False
    return True


class _SepPush(torch.autograd.Function):
    """The adjoint.  Round 5: D adjoint passes that GATHER (interpol_resample_1d, adjoint: for a non-decreasing `lin` the samples
    whose stencil covers a lattice point are a contiguous range -- no atomics, no zero-fill, K + 1 taps per pass), first dim first:
    the tensor shrinks before the pass whose lanes run along the lattice (4 x 2 x 256^3 -> 128^3: see profiles/r05_other_configs.json,
    f2).  Else ONE D-dimensional push on the separable lattice (the LDS-tiled scatter reads the D lattice vectors: 0.92 ms linear /
    2.27 cubic for that shape; D scattering passes with atomics 3.0 ms).  backward = `_SepPull`."""

    @staticmethod
    def forward(ctx, x, lin, shape, orders, bounds, extrapolate):
        D = len(lin)
        ctx.args = (lin, orders, bounds, extrapolate)
        if _gathers(x, lin) and all(int(x.shape[d - D]) == lin[d].numel() for d in range(D)):
            mode, out = _mode(orders), x
            for d in range(D):
                out = ops.resample1d(out, lin[d], d - D, orders[d], bounds[d], extrapolate, mode, adjoint=True, n_lattice=int(shape[d]))
            return out
        lead = x.shape[:-D]
        xf = x.reshape(-1, 1, *x.shape[-D:]) if len(lead) != 2 else x
        out = ops.grid_push(xf, SeparableGrid(lin), list(shape), bounds, orders, extrapolate)
        return out.reshape(*lead, *out.shape[-D:])

    @staticmethod
    def backward(ctx, grad):
        lin, orders, bounds, extrapolate = ctx.args
        gx = _SepPull.apply(grad, lin, orders, bounds, extrapolate) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, None, None


def _codes(lin, interpolation, bound):
    dim = len(lin)
    b = pad_codes([bound_to_code(x) for x in _as_list(bound)], dim)
    o = pad_codes([order_to_code(x) for x in _as_list(interpolation)], dim)
    return b, o


def separable_pull(image, lin, interpolation, bound, extrapolate):
    """image (..., *inshape), lin = D coordinate vectors -> (..., *[len(l) for l in lin]);
    equals grid_pull(image, stack(meshgrid_ij(*lin), -1), ...)."""
    b, o = _codes(lin, interpolation, bound)
    return _SepPull.apply(image, [l.detach() for l in lin], o, b, int(extrapolate))


def separable_push(image, lin, shape, interpolation, bound, extrapolate):
    """image (..., *[len(l) for l in lin]) -> (..., *shape); equals
    grid_push(image, stack(meshgrid_ij(*lin), -1), shape, ...)."""
    b, o = _codes(lin, interpolation, bound)
    return _SepPush.apply(image, [l.detach() for l in lin], [int(n) for n in shape], o, b, int(extrapolate))
