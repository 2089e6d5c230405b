"""Pins the CPU oracle against the live reference (build container only).

The reference is imported from /root/reference under the alias `interpol_ref`;
on machines without it (the GPU box) these tests skip and the committed golden
vectors (tests/test_oracle_golden.py) pin the oracle instead.
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.reference

SHAPES_IN = (5, 6, 7)
SHAPES_OUT = (4, 3, 5)


def make_case(dim, dtype, seed=0, B=2, C=3):
    g = torch.Generator().manual_seed(seed)
    ishape = SHAPES_IN[:dim]
    oshape = SHAPES_OUT[:dim]
    inp = torch.randn([B, C, *ishape], generator=g, dtype=torch.float64)
    lin = [torch.linspace(-1.0, n, m, dtype=torch.float64) for n, m in zip(ishape, oshape)]
    grid = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None].repeat(B, *([1] * (dim + 1)))
    grid = grid + 1.5 * torch.randn(grid.shape, generator=g, dtype=torch.float64)
    flat = grid.reshape(B, -1, dim)
    # far out-of-bounds samples, exact integers, exact halves
    flat[0, 0] = -3.0 * torch.tensor(ishape, dtype=torch.float64)
    flat[0, 1] = 3.0 * torch.tensor(ishape, dtype=torch.float64) + 0.25
    flat[1, 0] = 2.0
    flat[1, 1] = 1.5
    flat[1, 2] = 0.5
    flat[0, 2] = -0.5
    return inp.to(dtype), grid.to(dtype)


def close(a, b, dtype):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    err = float((a - b).abs().max()) / scale
    assert err <= tol, err


CASES = [(k, b) for k in range(8) for b in range(7)]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("extrapolate", [1, 0, 2])
def test_isotropic_sweep(reference, dim, dtype, extrapolate):
    from interpol_ref import pushpull as P
    inp, grid = make_case(dim, dtype, seed=dim)
    gin = grid  # push: grid must have input's spatial shape
    for order, bound in CASES:
        if extrapolate != 1 and (order + bound) % 3 != 0:
            continue  # thin the masked sweeps
        b, o = [bound], [order]
        skip_b1 = (order == 0 and dim == 2 and extrapolate != 1)   # reference bug B-1
        if not skip_b1:
            close(oracle.grid_pull(inp, grid, b, o, extrapolate), P.grid_pull(inp, grid, b, o, extrapolate), dtype)
        close(oracle.grid_grad(inp, grid, b, o, extrapolate), P.grid_grad(inp, grid, b, o, extrapolate), dtype)
        val = inp[..., :1].expand(*inp.shape[:2], *grid.shape[1:-1]) if False else None
        src = torch.randn([inp.shape[0], inp.shape[1], *grid.shape[1:-1]],
                          generator=torch.Generator().manual_seed(7), dtype=torch.float64).to(dtype)
        shape = list(inp.shape[2:])
        close(oracle.grid_push(src, gin, shape, b, o, extrapolate), P.grid_push(src, gin, shape, b, o, extrapolate), dtype)
        close(oracle.grid_count(gin, shape, b, o, extrapolate), P.grid_count(gin, shape, b, o, extrapolate), dtype)
        if extrapolate == 1:
            src4 = torch.randn([*src.shape, dim], generator=torch.Generator().manual_seed(8),
                               dtype=torch.float64).to(dtype)
            close(oracle.grid_pushgrad(src4, gin, shape, b, o, extrapolate),
                  P.grid_pushgrad(src4, gin, shape, b, o, extrapolate), dtype)
            close(oracle.grid_hess(inp, grid, b, o, extrapolate), P.grid_hess(inp, grid, b, o, extrapolate), dtype)


MIXED = [
    ([2, 3, 5], [2, 5, 0]),
    ([1, 3], [6, 1, 3]),
    ([0, 3], [3]),
    ([3, 1, 2], [4, 2, 6]),
    ([1, 1, 0], [0, 4, 5]),
    ([7, 0, 4], [1, 1, 1]),
]


@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("orders,bounds", MIXED)
def test_mixed_orders_and_bounds(reference, dim, orders, bounds):
    from interpol_ref import pushpull as P
    dtype = torch.float64
    inp, grid = make_case(dim, dtype, seed=10 + dim)
    for ex in (1, 0):
        close(oracle.grid_pull(inp, grid, bounds, orders, ex), P.grid_pull(inp, grid, bounds, orders, ex), dtype)
        close(oracle.grid_grad(inp, grid, bounds, orders, ex), P.grid_grad(inp, grid, bounds, orders, ex), dtype)
        src = torch.randn([inp.shape[0], inp.shape[1], *grid.shape[1:-1]], dtype=dtype,
                          generator=torch.Generator().manual_seed(3))
        shape = list(inp.shape[2:])
        close(oracle.grid_push(src, grid, shape, bounds, orders, ex), P.grid_push(src, grid, shape, bounds, orders, ex), dtype)
    close(oracle.grid_hess(inp, grid, bounds, orders, 1), P.grid_hess(inp, grid, bounds, orders, 1), dtype)


def test_batch_broadcast(reference):
    from interpol_ref import pushpull as P
    inp, grid = make_case(2, torch.float64, seed=5)
    # (the reference operators only broadcast the *grid* batch: nd.py:121-123 expands idx, not inp)
    close(oracle.grid_pull(inp, grid[:1], [3], [3], 1), P.grid_pull(inp, grid[:1], [3], [3], 1), torch.float64)
    close(oracle.grid_grad(inp, grid[:1], [6], [2], 1), P.grid_grad(inp, grid[:1], [6], [2], 1), torch.float64)
    src = torch.randn([2, 3, *grid.shape[1:-1]], dtype=torch.float64)
    close(oracle.grid_push(src, grid[:1], [5, 6], [3], [3], 1), P.grid_push(src, grid[:1], [5, 6], [3], [3], 1), torch.float64)


def test_backward_compositions(reference):
    from interpol_ref import pushpull as P
    inp, grid = make_case(3, torch.float64, seed=9)
    inp.requires_grad_(True)
    grid.requires_grad_(True)
    gout = torch.randn([2, 3, *grid.shape[1:-1]], dtype=torch.float64)
    torch.set_grad_enabled(False)   # the reference's *_backward run inside autograd's no-grad backward
    for b, o in (([3], [3]), ([6], [5]), ([1], [1]), ([2, 5, 0], [2, 3, 1])):
        a = oracle.grid_pull_backward(gout, inp, grid, b, o, 1)
        r = P.grid_pull_backward(gout, inp, grid, b, o, 1)
        close(a[0], r[0].detach(), torch.float64)
        close(a[1], r[1].detach(), torch.float64)
    src = torch.randn([2, 3, *grid.shape[1:-1]], dtype=torch.float64, requires_grad=True)
    gvol = torch.randn([2, 3, 5, 6, 7], dtype=torch.float64)
    a = oracle.grid_push_backward(gvol, src, grid, [3], [3], 1)
    r = P.grid_push_backward(gvol, src, grid, [3], [3], 1)
    close(a[0], r[0].detach(), torch.float64)
    close(a[1], r[1].detach(), torch.float64)
    a = oracle.grid_count_backward(gvol[:, :1], grid, [3], [3], 1)
    r = P.grid_count_backward(gvol[:, :1], grid, [3], [3], 1)
    close(a, r.detach(), torch.float64)
    gg = torch.randn([2, 3, *grid.shape[1:-1], 3], dtype=torch.float64)
    a = oracle.grid_grad_backward(gg, inp, grid, [3], [3], 1)
    r = P.grid_grad_backward(gg, inp, grid, [3], [3], 1)
    close(a[0], r[0].detach(), torch.float64)
    close(a[1], r[1].detach(), torch.float64)
    torch.set_grad_enabled(True)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_prefilter(reference, dtype):
    from interpol_ref import coeff as K
    g = torch.Generator().manual_seed(42)
    for n in (1, 2, 3, 7, 9, 11, 64, 200):
        x = torch.randn([3, n, 4], generator=g, dtype=torch.float64).to(dtype)
        for order in range(0, 8):
            for bound in (0, 1, 2, 3, 6):
                ref = K.spline_coeff(x, bound, order, dim=1)
                got = oracle.spline_coeff(x, bound, order, dim=1)
                a, b = got.double(), ref.double()
                tol = 1e-11 if dtype == torch.float64 else 5e-5
                assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (n, order, bound)
    x = torch.randn([2, 9, 10, 11], generator=g, dtype=torch.float64)
    ref = K.spline_coeff_nd(x, [2, 3, 6], [2, 3, 5], 3)
    got = oracle.spline_coeff_nd(x, [2, 3, 6], [2, 3, 5], 3)
    assert float((got - ref).abs().max()) < 1e-11
    with pytest.raises(NotImplementedError):
        oracle.spline_coeff(x, 5, 3, dim=-1)


def test_index_and_sign_tables(reference):
    from interpol_ref.bounds import Bound
    for n in (1, 2, 3, 4, 7):
        i = torch.arange(-4 * n - 3, 4 * n + 4)
        for b in range(7):
            bb = Bound(b)
            idx = bb.index(i, n).tolist()
            sgn = bb.transform(i, n)
            for ii, want in zip(i.tolist(), idx):
                assert oracle.bound_index(b, ii, n) == want, (b, n, ii)
            for k, ii in enumerate(i.tolist()):
                got = oracle.bound_sign(b, ii, n)
                if sgn is None:
                    assert got is None
                else:
                    assert got == int(sgn[k]), (b, n, ii)


def test_weights_bitexact_fp32(reference):
    from interpol_ref.splines import Spline
    for k in range(8):
        s = Spline(k)
        half = (k + 1) / 2
        x = torch.linspace(-half, half, 4001, dtype=torch.float32)
        w, g, h = s.fastweight(x), s.fastgrad(x), s.fasthess(x)
        for j in range(0, len(x), 7):
            xv = float(x[j])
            assert oracle.weight(k, xv, "f32") == float(w[j]), (k, xv)
            assert oracle.wgrad(k, xv, "f32") == float(g[j]), (k, xv)
            assert oracle.whess(k, xv, "f32") == float(h[j]), (k, xv)
