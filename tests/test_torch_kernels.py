"""The device-generic PyTorch kernel table (interpol/torch_kernels.py, interpol/filter_torch.py): what serves CPU tensors
and grids of more than three spatial dims.  Checked on the CPU against the oracle (D <= 3: every operator, order, bound,
extrapolation mode, the backward compositions, the prefilter) and against golden vectors generated from the reference for
D = 4 (tests/golden/make_golden_nd4.py).  fp64: 1e-10 of the largest value."""
import os

import numpy as np
import pytest
import torch

import interpol
from interpol import ops
from interpol.torch_kernels import TorchKernels, bound_index, bound_sign
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _close(a, r, tol, what):
    a = a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a)
    r = np.asarray(r, dtype=np.float64)
    assert a.shape == r.shape, (what, a.shape, r.shape)
    err = np.abs(a - r).max() if a.size else 0.0
    assert err <= tol * max(np.abs(r).max() if r.size else 0.0, 1e-30), (what, err)


def _problem(dim, seed, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    ishape, oshape = (5, 6, 7)[:dim], (4, 3, 5)[:dim]
    inp = torch.randn([2, 3, *ishape], generator=gen, dtype=dtype)
    scale = (torch.tensor(ishape, dtype=dtype) - 1) / (torch.tensor(oshape, dtype=dtype) - 1)
    grid = (interpol.identity_grid(oshape, dtype=dtype) * scale)[None] + 1.5 * torch.randn([2, *oshape, dim], generator=gen, dtype=dtype)
    grid[0].reshape(-1, dim)[0] = -3.0 * max(ishape)                 # far outside
    grid[1].reshape(-1, dim)[1] = 3.0 * max(ishape)
    grid[0].reshape(-1, dim)[2] = 1.0                                # exact integer / exact half coordinates
    grid[1].reshape(-1, dim)[2] = 1.5
    src = torch.randn([2, 3, *oshape], generator=gen, dtype=dtype)
    return inp, grid, src, list(ishape)


def test_bound_tables_match_the_oracle():
    i = torch.arange(-70, 70)
    for b in range(7):
        for n in (1, 2, 5, 13):
            idx = bound_index(b, i, n)
            sg = bound_sign(b, i, n)
            for k, ii in enumerate(i.tolist()):
                assert int(idx[k]) == oracle.bound_index(b, ii, n), (b, n, ii)
                want = oracle.bound_sign(b, ii, n)
                got = 1 if sg is None else int(sg[k])
                assert got == (1 if want in (None, 2) else want), (b, n, ii)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_all_operators_against_the_oracle(dim):
    for order in range(8):
        for bound in range(7):
            for ex in (0, 1, 2):
                if (order + bound + ex) % 3 and dim == 3:           # (a third of the 3-D sweep: seconds, not minutes)
                    continue
                inp, grid, src, ishape = _problem(dim, 100 * dim + 10 * order + bound)
                o, b = [order] * dim, [bound] * dim
                what = (dim, order, bound, ex)
                _close(TorchKernels.pull(inp, grid, b, o, ex), oracle.grid_pull(inp, grid, b, o, ex), 1e-10, ("pull",) + what)
                _close(TorchKernels.push(src, grid, ishape, b, o, ex), oracle.grid_push(src, grid, ishape, b, o, ex), 1e-10, ("push",) + what)
                _close(TorchKernels.count(grid, ishape, b, o, ex), oracle.grid_count(grid, ishape, b, o, ex), 1e-10, ("count",) + what)
                _close(TorchKernels.grad(inp, grid, b, o, ex), oracle.grid_grad(inp, grid, b, o, ex), 1e-10, ("grad",) + what)
                if ex == 1:
                    _close(TorchKernels.hess(inp, grid, b, o, ex), oracle.grid_hess(inp, grid, b, o, ex), 1e-9, ("hess",) + what)
                    gsrc = torch.randn(list(src.shape) + [dim], dtype=torch.float64, generator=torch.Generator().manual_seed(1))
                    _close(TorchKernels.pushgrad(gsrc, grid, ishape, b, o, ex), oracle.grid_pushgrad(gsrc, grid, ishape, b, o, ex), 1e-10,
                           ("pushgrad",) + what)


def test_mixed_orders_bounds_and_backward():
    for dim, order, bound in [(2, [2, 3], [2, 5]), (3, [1, 3, 2], [6, 1, 3]), (3, [0, 3, 1], [4, 0, 3]), (2, [1, 1], [0, 0]), (3, [0, 0, 0], [3, 3, 3])]:
        inp, grid, src, ishape = _problem(dim, 7 * dim + order[0])
        for ex in (0, 1):
            what = (dim, order, bound, ex)
            _close(TorchKernels.pull(inp, grid, bound, order, ex), oracle.grid_pull(inp, grid, bound, order, ex), 1e-10, ("pull",) + what)
            _close(TorchKernels.grad(inp, grid, bound, order, ex), oracle.grid_grad(inp, grid, bound, order, ex), 1e-10, ("grad",) + what)
            gi, gg = TorchKernels.pull_backward(src, inp, grid, bound, order, ex, True, True)
            wi, wg = oracle.grid_pull_backward(src, inp, grid, bound, order, ex)
            _close(gi, wi, 1e-10, ("pull bwd inp",) + what); _close(gg, wg, 1e-10, ("pull bwd grid",) + what)
            gvo = torch.randn([2, 3, *ishape], dtype=torch.float64, generator=torch.Generator().manual_seed(3))
            gi, gg = TorchKernels.push_backward(gvo, src, grid, bound, order, ex, True, True)
            wi, wg = oracle.grid_push_backward(gvo, src, grid, bound, order, ex)
            _close(gi, wi, 1e-10, ("push bwd inp",) + what); _close(gg, wg, 1e-10, ("push bwd grid",) + what)
            _close(TorchKernels.count_backward(gvo[:, :1], grid, bound, order, ex), oracle.grid_count_backward(gvo[:, :1], grid, bound, order, ex),
                   1e-10, ("count bwd",) + what)


def test_four_dimensional_grids_against_the_reference_vectors():
    z = np.load(os.path.join(HERE, "golden", "golden_nd4.npz"))
    for k, row in enumerate(z["cases"]):
        order, bound, ex = row[:4].tolist(), row[4:8].tolist(), int(row[8])
        pre = "c%d_" % k
        inp, grid, src = (torch.from_numpy(z[pre + n]) for n in ("inp", "grid", "src"))
        ishape = list(inp.shape[2:])
        _close(TorchKernels.pull(inp, grid, bound, order, ex), z[pre + "pull"], 1e-10, ("pull", k))
        _close(TorchKernels.push(src, grid, ishape, bound, order, ex), z[pre + "push"], 1e-10, ("push", k))
        _close(TorchKernels.count(grid, ishape, bound, order, ex), z[pre + "count"], 1e-10, ("count", k))
        _close(TorchKernels.grad(inp, grid, bound, order, ex), z[pre + "grad"], 1e-10, ("grad", k))


def test_api_serves_cpu_tensors_and_4d_grids_through_the_torch_table():
    """No test hook installed: CPU tensors / D > 3 reach torch_kernels.py through the public API, with autograd."""
    inp, grid, src, ishape = _problem(3, 55, torch.float32)
    inp = inp.requires_grad_(True); grid = grid.requires_grad_(True)
    y = interpol.grid_pull(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
    _close(y, oracle.grid_pull(inp.detach().double(), grid.detach().double(), [3], [3], 1), 2e-6, "api pull (cpu)")
    y.square().sum().backward()
    wi, wg = oracle.grid_pull_backward(2 * y.detach().double(), inp.detach().double(), grid.detach().double(), [3], [3], 1)
    _close(inp.grad, wi, 1e-5, "api pull grad_input (cpu)"); _close(grid.grad, wg, 1e-5, "api pull grad_grid (cpu)")
    z = interpol.grid_push(src, grid.detach(), ishape, interpolation=2, bound="replicate", extrapolate=False)
    _close(z, oracle.grid_push(src.double(), grid.detach().double(), ishape, [1], [2], 0), 2e-6, "api push (cpu)")
    c = interpol.spline_coeff_nd(inp.detach(), interpolation=3, bound="dct2", dim=3)
    _close(c, oracle.spline_coeff_nd(inp.detach().double().numpy(), [3], [3], 3), 1e-5, "api spline_coeff_nd (cpu)")
    r = interpol.resize(inp.detach(), factor=[2, 2, 2], interpolation=1, prefilter=False)
    assert list(r.shape[2:]) == [2 * n for n in ishape]
    lab = interpol.grid_pull(torch.randint(0, 4, [1, 1, 6, 6]), interpol.identity_grid([6, 6])[None] + 0.3, interpolation=1)
    assert lab.dtype == torch.int64
    z4 = np.load(os.path.join(HERE, "golden", "golden_nd4.npz"))
    got = interpol.grid_pull(torch.from_numpy(z4["c1_inp"]), torch.from_numpy(z4["c1_grid"]), interpolation=3, bound="dct2", extrapolate=True)
    _close(got, z4["c1_pull"], 1e-10, "api pull, D = 4")


@pytest.mark.parametrize("bound", [0, 1, 2, 3, 6])
def test_prefilter_against_the_oracle(bound):
    from interpol.filter_torch import spline_filter_
    gen = torch.Generator().manual_seed(9)
    for order in range(2, 8):
        for n in (1, 2, 3, 7, 11, 64):
            x = torch.randn([3, n, 4], generator=gen, dtype=torch.float64)
            got = spline_filter_(x.clone(), bound, order, 1)
            _close(got, oracle.spline_coeff(x.numpy(), bound, order, dim=1), 1e-10, ("prefilter", bound, order, n))
