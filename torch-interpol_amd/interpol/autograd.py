"""Differentiable operators (autograd boundary).

Counterpart of the reference's `interpol/autograd.py:157-333`: six
`torch.autograd.Function`s with the same `apply(...)` signatures, the same
alias handling (`bound_to_nitorch` / `inter_to_nitorch`), float32 up-casting
under CUDA autocast, and `requires_grad`-driven skipping in backward.  The
forward and backward bodies call the fused HIP operators of `ops.py`.
"""
import torch

from . import ops
from .coeff import _spline_coeff, _spline_coeff_nd
from .codes import bound_to_code, order_to_code
from .sepgrid import SeparableGrid, AffineGrid, LazyGrid

try:                                    # torch >= 2.4
    from torch.amp import custom_fwd, custom_bwd
    _fwd32 = custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    _fwd = custom_fwd(device_type='cuda')
    _bwd = custom_bwd(device_type='cuda')
except ImportError:                     # pragma: no cover
    from torch.cuda.amp import custom_fwd, custom_bwd
    _fwd32 = custom_fwd(cast_inputs=torch.float32)
    _fwd = custom_fwd
    _bwd = custom_bwd


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _save(ctx, input, grid):
    """save_for_backward with a constant SeparableGrid kept as a plain attribute."""
    if isinstance(grid, LazyGrid):
        ctx.sep = grid
        ctx.save_for_backward(input)
    else:
        ctx.sep = None
        ctx.save_for_backward(input, grid)


def _saved(ctx):
    if ctx.sep is not None:
        return ctx.saved_tensors[0], ctx.sep
    return ctx.saved_tensors


def _extra(displacement):
    """Optional trailing `displacement` argument of the sampling Functions (an extension: the
    reference's `apply` signatures stay valid).  -> (flag, number of extra inputs)."""
    return (bool(displacement[0]) if displacement else False), len(displacement)


def _higher_order():
    """Inside a `backward`: is a graph being recorded (create_graph=True)?  Then the gradients must
    themselves be differentiable: the backward is composed from the Functions of this module (as the
    reference's backward is composed from differentiable torch ops, pushpull.py:237-325) instead of
    calling the fused kernels, whose outputs carry no graph."""
    return torch.is_grad_enabled()


def _extra_args(ctx):
    return (ctx.disp,) if ctx.nextra else ()


def _options(bound, interpolation, extrapolate):
    return ([bound_to_code(b) for b in _as_list(bound)],
            [order_to_code(o) for o in _as_list(interpolation)],
            int(extrapolate))


class GridPull(torch.autograd.Function):
    """reference interpol/autograd.py:157-184"""

    @staticmethod
    @_fwd32
    def forward(ctx, input, grid, interpolation, bound, extrapolate, *displacement):
        opt = _options(bound, interpolation, extrapolate)
        ctx.disp, ctx.nextra = _extra(displacement)
        output = ops.grid_pull(input, grid, *opt, displacement=ctx.disp)
        ctx.opt = opt
        _save(ctx, input, grid)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        input, grid = _saved(ctx)
        if _higher_order():
            # pull backward = push of the gradient & contraction of grid_grad (pushpull.py:237-258)
            bound, interpolation, extrapolate = ctx.opt
            grad_input = grad_grid = None
            if ctx.needs_input_grad[0]:
                grad_input = GridPush.apply(grad, grid, list(input.shape[2:]), interpolation, bound, extrapolate, *_extra_args(ctx))
            if ctx.needs_input_grad[1]:
                grad_grid = (GridGrad.apply(input, grid, interpolation, bound, extrapolate, *_extra_args(ctx)) * grad.unsqueeze(-1)).sum(1)
            return (grad_input, grad_grid, None, None, None) + (None,) * ctx.nextra
        grad_input, grad_grid = ops.grid_pull_backward(
            grad, input, grid, *ctx.opt,
            need_inp=ctx.needs_input_grad[0], need_grid=ctx.needs_input_grad[1], displacement=ctx.disp)
        return (grad_input, grad_grid, None, None, None) + (None,) * ctx.nextra


class GridPush(torch.autograd.Function):
    """reference interpol/autograd.py:187-214"""

    @staticmethod
    @_fwd32
    def forward(ctx, input, grid, shape, interpolation, bound, extrapolate, *displacement):
        opt = _options(bound, interpolation, extrapolate)
        ctx.disp, ctx.nextra = _extra(displacement)
        output = ops.grid_push(input, grid, shape, *opt, displacement=ctx.disp)
        ctx.opt = opt
        _save(ctx, input, grid)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        input, grid = _saved(ctx)
        if _higher_order():
            # push backward = pull of the gradient & contraction of its grid_grad (pushpull.py:262-292)
            bound, interpolation, extrapolate = ctx.opt
            grad_input = grad_grid = None
            if ctx.needs_input_grad[0]:
                grad_input = GridPull.apply(grad, grid, interpolation, bound, extrapolate, *_extra_args(ctx))
            if ctx.needs_input_grad[1]:
                grad_grid = (GridGrad.apply(grad, grid, interpolation, bound, extrapolate, *_extra_args(ctx)) * input.unsqueeze(-1)).sum(1)
            return (grad_input, grad_grid, None, None, None, None) + (None,) * ctx.nextra
        grad_input, grad_grid = ops.grid_push_backward(
            grad, input, grid, *ctx.opt,
            need_inp=ctx.needs_input_grad[0], need_grid=ctx.needs_input_grad[1], displacement=ctx.disp)
        return (grad_input, grad_grid, None, None, None, None) + (None,) * ctx.nextra


class GridCount(torch.autograd.Function):
    """reference interpol/autograd.py:217-245"""

    @staticmethod
    @_fwd32
    def forward(ctx, grid, shape, interpolation, bound, extrapolate, *displacement):
        opt = _options(bound, interpolation, extrapolate)
        ctx.disp, ctx.nextra = _extra(displacement)
        output = ops.grid_count(grid, shape, *opt, displacement=ctx.disp)
        ctx.opt = opt
        ctx.save_for_backward(grid)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        grid, = ctx.saved_tensors
        grad_grid = None
        if ctx.needs_input_grad[0] and _higher_order():
            # count backward = grid_grad of the gradient, summed over its channel (pushpull.py:296-325)
            bound, interpolation, extrapolate = ctx.opt
            grad_grid = GridGrad.apply(grad, grid, interpolation, bound, extrapolate, *_extra_args(ctx)).sum(1)
        elif ctx.needs_input_grad[0]:
            grad_grid = ops.grid_count_backward(grad, grid, *ctx.opt, need_grid=True, displacement=ctx.disp)
        return (grad_grid, None, None, None, None) + (None,) * ctx.nextra


class GridGrad(torch.autograd.Function):
    """reference interpol/autograd.py:248-277"""

    @staticmethod
    @_fwd32
    def forward(ctx, input, grid, interpolation, bound, extrapolate, *displacement):
        opt = _options(bound, interpolation, extrapolate)
        ctx.disp, ctx.nextra = _extra(displacement)
        output = ops.grid_grad(input, grid, *opt, displacement=ctx.disp)
        ctx.opt = opt
        _save(ctx, input, grid)
        return output

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        input, grid = _saved(ctx)
        grad_input = grad_grid = None
        if _higher_order() and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            # Third order and beyond (create_graph=True inside a double backward).  The reference's backward is plain torch and
            # differentiates on (pushpull.py:303-325); the fused kernels (pushgrad, hess) carry no graph, so here autograd itself
            # differentiates the differentiable restatement of grid_grad (torch_kernels.py: weights and their derivatives as torch
            # expressions of the coordinates, one gather per chunk) -- every order from here on, at PyTorch speed.
            from .torch_kernels import TorchKernels
            bound, interpolation, extrapolate = ctx.opt
            bound, interpolation = ops._codes(grid, bound, interpolation)
            wrt = [t for t, need in ((input, ctx.needs_input_grad[0]), (grid, ctx.needs_input_grad[1])) if need]
            with torch.enable_grad():
                out = TorchKernels.grad(input, grid, bound, interpolation, int(extrapolate), displacement=ctx.disp)
                got = list(torch.autograd.grad(out, wrt, grad.to(out.dtype), create_graph=True, allow_unused=True))
            if ctx.needs_input_grad[0]:
                grad_input = got.pop(0)
            if ctx.needs_input_grad[1]:
                grad_grid = got.pop(0)
            return (grad_input, grad_grid, None, None, None) + (None,) * ctx.nextra
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_grid = ops.grid_grad_backward(
                grad, input, grid, *ctx.opt,
                need_inp=ctx.needs_input_grad[0], need_grid=ctx.needs_input_grad[1], displacement=ctx.disp)
        return (grad_input, grad_grid, None, None, None) + (None,) * ctx.nextra


class SplineCoeff(torch.autograd.Function):
    """reference interpol/autograd.py:280-305"""

    @staticmethod
    @_fwd
    def forward(ctx, input, bound, interpolation, dim, inplace):
        bound = bound_to_code(_as_list(bound)[0])
        interpolation = order_to_code(_as_list(interpolation)[0])
        ctx.opt = (bound, interpolation, dim)
        return _spline_coeff(input, bound, interpolation, dim, inplace)

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        # the filter is symmetric: backward == forward (autograd.py:300-305)
        if _higher_order():
            return SplineCoeff.apply(grad, *ctx.opt, False), None, None, None, None
        return _spline_coeff(grad, *ctx.opt, inplace=False), None, None, None, None


class SplineCoeffND(torch.autograd.Function):
    """reference interpol/autograd.py:308-333"""

    @staticmethod
    @_fwd
    def forward(ctx, input, bound, interpolation, dim, inplace):
        bound = [bound_to_code(b) for b in _as_list(bound)]
        interpolation = [order_to_code(o) for o in _as_list(interpolation)]
        ctx.opt = (bound, interpolation, dim)
        return _spline_coeff_nd(input, bound, interpolation, dim, inplace)

    @staticmethod
    @_bwd
    def backward(ctx, grad):
        if _higher_order():
            return SplineCoeffND.apply(grad, *ctx.opt, False), None, None, None, None
        return _spline_coeff_nd(grad, *ctx.opt, inplace=False), None, None, None, None
