"""Interpolating-spline prefilter (B-spline coefficients).

Counterpart of the reference's `interpol/coeff.py:288-347`.  The recursion the
reference runs as a Python loop of n tensor ops per pole (coeff.py:272-281) is
one HIP kernel per filtered dimension (`csrc/prefilter.hip`), all poles fused.
"""
from . import ops
from .codes import pad_codes


def _spline_coeff(inp, bound, order, dim=-1, inplace=False):
    """coeff.py:288-313 (int codes).  dst1/dst2 raise NotImplementedError like the
    reference (coeff.py:243-244)."""
    if order > 7:
        raise NotImplementedError
    if order in (0, 1):
        return inp if inplace else inp.clone()
    if bound in (4, 5):
        raise NotImplementedError('spline prefilter is not implemented for dst1/dst2 boundaries')
    if inp.dim() == 0:
        return inp if inplace else inp.clone()
    if inplace and inp.is_contiguous():
        out = ops.kernels(inp).spline_filter_(inp, bound, order, dim)
    elif inp.is_contiguous():
        # out of place: the kernel reads `inp` and writes the result (no copy first)
        out = ops.kernels(inp).spline_filter_(inp.new_empty(inp.shape), bound, order, dim, src=inp)
    else:
        out = ops.kernels(inp).spline_filter_(inp.contiguous(), bound, order, dim)
    if inplace and out is not inp:
        inp.copy_(out)
        return inp
    return out


def _spline_coeff_nd(inp, bound, order, dim=None, inplace=False):
    """coeff.py:317-347: filter the last `dim` dimensions (ALL dimensions,
    batch included, when `dim` is None -- coeff.py:338-339)."""
    if dim is None:
        dim = inp.dim()
    if dim == 0:
        return inp if inplace else inp.clone()
    bound = pad_codes(bound, dim)
    order = pad_codes(order, dim)
    if any(o > 7 for o in order):
        raise NotImplementedError
    if any(b in (4, 5) and o > 1 for b, o in zip(bound, order)):
        raise NotImplementedError('spline prefilter is not implemented for dst1/dst2 boundaries')
    todo = [(d, b, o) for d, (b, o) in enumerate(zip(bound, order)) if o > 1]
    if inplace and inp.is_contiguous():
        out, src = inp, None
    elif inp.is_contiguous() and todo:
        out, src = inp.new_empty(inp.shape), inp        # the first pass reads `inp` and writes `out` (no copy first)
    else:
        out, src = (inp.contiguous() if not inp.is_contiguous() else inp.clone()), None
    for d, b, o in todo:
        ops.kernels(out).spline_filter_(out, b, o, -dim + d, src=src)
        src = None
    if inplace and out is not inp:
        inp.copy_(out)
        return inp
    return out
