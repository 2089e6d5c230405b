"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared
symbol, its host-side scalar primitives match the reference tables, and the
Python host logic (shape conventions, aliases, errors, autograd wiring) behaves
like the reference's api.py / autograd.py.  Compute on CPU goes through the
TEST-ONLY oracle kernel table; the product path itself refuses CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_util as G
import interpol
from interpol import _hip, ops
from interpol.codes import bound_to_nitorch, inter_to_nitorch
from oracle import oracle
from oracle_kernels import OracleKernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "interpol_hip.h")).read()
    declared = set(re.findall(r"\b(interpol_[a-z0-9_]+)\s*\(", header))
    declared -= {"interpol_problem"}
    assert declared == set(_hip.SYMBOLS), declared ^ set(_hip.SYMBOLS)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _hip.lib().interpol_abi_version() == 1
    assert ctypes.sizeof(_hip.Problem) == 4 * 12 + 8 * 2 + 8 * 6 + 8 * 17


def test_host_index_and_sign_tables():
    L = _hip.lib()
    for key, t in G.api()["tables"].items():
        n, b = int(key[1:key.index("_")]), int(key[-1])
        for k, want in enumerate(t["idx"]):
            i = t["i0"] + k
            assert L.interpol_host_bound_index(b, i, n) == want, (b, n, i)
            s = L.interpol_host_bound_sign(b, i, n)
            assert s == (2 if t["sign"] is None else t["sign"][k]), (b, n, i)
    # far outside (modulo path) against the oracle
    for b in range(7):
        for n in (1, 2, 5, 13):
            for i in list(range(-70, 70)) + [-100003, 99991]:
                assert L.interpol_host_bound_index(b, i, n) == oracle.bound_index(b, i, n), (b, n, i)
                s = oracle.bound_sign(b, i, n)
                assert L.interpol_host_bound_sign(b, i, n) == (2 if s is None else s), (b, n, i)


def test_host_weights_match_oracle():
    L = _hip.lib()
    for k in range(8):
        half = (k + 1) / 2
        for x in np.linspace(-half, half, 401):
            for which, fn in enumerate((oracle.weight, oracle.wgrad, oracle.whess)):
                got = L.interpol_host_weight(k, float(x), which)
                assert abs(got - fn(k, float(x))) < 1e-13, (k, x, which)
            got32 = L.interpol_host_weight_f32(k, float(x), 0)
            # the fp32 weights follow the EXACT spline (middle pieces expanded about their outer breakpoints, csrc/spline_math.hpp);
            # the reference's own fp32 Horner forms (the oracle's 'f32' mode) cancel near the knots and stray up to 1e-5
            assert abs(got32 - oracle.weight(k, float(np.float32(x)))) < 3e-7, (k, x)
            assert abs(got32 - oracle.weight(k, float(np.float32(x)), 'f32')) < 1.5e-5
    # partition of unity
    for k in range(8):
        for f in np.linspace(0, 0.999, 37):
            t = f + (k - 1) / 2 if k % 2 else f + (k - 1) / 2
            s = sum(L.interpol_host_weight(k, t - j, 0) for j in range(k + 1))
            assert abs(s - 1) < 1e-14


def test_alias_tables():
    api = G.api()
    for k, v in api["bounds"].items():
        assert bound_to_nitorch(int(k) if k.isdigit() else k, "int") == v
    for k, v in api["interpolations"].items():
        assert inter_to_nitorch(int(k) if k.isdigit() else k, "int") == v
    assert bound_to_nitorch(["reflect", 6], "str") == ["dct2", "dft"]
    assert inter_to_nitorch(("cubic", 1), "int") == (3, 1)
    with pytest.raises(ValueError, match="Unknown boundary condition"):
        bound_to_nitorch("bogus")
    with pytest.raises(ValueError, match="Unknown interpolation order"):
        inter_to_nitorch(8)
    with pytest.raises(ValueError):
        bound_to_nitorch(7)


def test_hip_layer_refuses_cpu_tensors_loudly():
    """The C-ABI layer has no CPU path and says so; the public API serves CPU tensors through the product's own PyTorch
    restatement instead (interpol/torch_kernels.py, tests/test_torch_kernels.py) -- never through a silent fallback of the
    GPU path: CUDA tensors of 1-3 spatial dims always go to libinterpol_hip.so (ops.kernels)."""
    x, g = torch.randn(1, 1, 4, 4), torch.rand(1, 4, 4, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _hip.gather("pull", x, g, [1, 1], [1, 1], 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _hip.scatter("push", x, g, None, [1, 1], [1, 1], 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _hip.scatter("count", None, g, None, [1, 1], [1, 1], 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _hip.spline_filter_(x.clone(), 3, 3, -1)
    from interpol.torch_kernels import TorchKernels
    assert ops.kernels(x, g, dim=2) is TorchKernels
    assert ops.kernels(dim=2) is ops._HipKernels and ops.kernels(dim=4) is TorchKernels


def test_api_shape_conventions():
    torch.manual_seed(0)
    with ops.use_kernels(OracleKernels):
        for c in G.api()["shapes"]:
            args = []
            if c["input"] is not None:
                args.append(torch.randn(c["input"], dtype=torch.float64))
            args.append(torch.rand(c["grid"], dtype=torch.float64) * 4)
            kw = dict(c["kwargs"])
            out = getattr(interpol, c["fn"])(*args, **kw)
            assert list(out.shape) == c["out"], c
        with pytest.raises(ValueError, match="Incompatible shapes for broadcasting"):
            interpol.grid_pull(torch.randn(3, 1, 5, 6), torch.rand(2, 5, 6, 2))
        with pytest.raises(ValueError, match="same spatial shape"):
            ops.grid_push(torch.randn(1, 1, 5, 6), torch.rand(1, 4, 6, 2), None, [0], [1], 1)


def test_list_padding_and_truncation():
    """3-element lists on a 2-D problem keep their first two entries (jit_utils.py:9-15)."""
    torch.manual_seed(1)
    x = torch.randn(2, 3, 7, 8, dtype=torch.float64)
    g = torch.rand(2, 5, 6, 2, dtype=torch.float64) * 9 - 1
    with ops.use_kernels(OracleKernels):
        a = interpol.grid_pull(x, g, interpolation=[2, 3, 5], bound=['dct1', 'dst2', 'zero'], extrapolate=True)
        b = interpol.grid_pull(x, g, interpolation=[2, 3], bound=['dct1', 'dst2'], extrapolate=True)
        c = interpol.grid_pull(x, g, interpolation=[3], bound='dct2', extrapolate=True)
        d = interpol.grid_pull(x, g, interpolation=[3, 3], bound=['dct2', 'dct2'], extrapolate=True)
    assert torch.equal(a, b) and torch.equal(c, d)


def _bwd_case(c, dtype=torch.float64):
    kw = dict(interpolation=c["interpolation"], bound=c["bound"], extrapolate=c["extrapolate"])
    grid = torch.from_numpy(G.arr(c["grid"])).to(dtype).requires_grad_(True)
    gout = torch.from_numpy(G.arr(c["gout"])).to(dtype)
    inp = None
    if c["fn"] == "grid_count":
        out = interpol.grid_count(grid, c["shape"], **kw)
    else:
        inp = torch.from_numpy(G.arr(c["inp"])).to(dtype).requires_grad_(True)
        if c["fn"] == "grid_push":
            out = interpol.grid_push(inp, grid, c["shape"], **kw)
        else:
            out = getattr(interpol, c["fn"])(inp, grid, **kw)
    out.backward(gout)
    return out, inp, grid


def test_autograd_wiring_against_reference_backward():
    with ops.use_kernels(OracleKernels):
        for c in G.manifest()["backward"]:
            out, inp, grid = _bwd_case(c)
            assert G.rel_err(out.detach().numpy(), G.arr(c["out"])) < 1e-12, c["fn"]
            if inp is not None:
                assert G.rel_err(inp.grad.numpy(), G.arr(c["grad_inp"])) < 1e-12, c
            assert G.rel_err(grid.grad.numpy(), G.arr(c["grad_grid"])) < 1e-12, c


def test_requires_grad_driven_skipping():
    calls = []

    class Spy(OracleKernels):
        @staticmethod
        def pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid):
            calls.append((need_inp, need_grid))
            return OracleKernels.pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid)

    x = torch.randn(1, 1, 5, 6, dtype=torch.float64)
    g = torch.rand(1, 5, 6, 2, dtype=torch.float64) * 4
    with ops.use_kernels(Spy):
        interpol.grid_pull(x.clone().requires_grad_(True), g, interpolation=3).sum().backward()
        interpol.grid_pull(x, g.clone().requires_grad_(True), interpolation=3).sum().backward()
    assert calls == [(True, False), (False, True)]


def test_prefilter_api_and_errors():
    torch.manual_seed(3)
    x = torch.randn(2, 9, 10, dtype=torch.float64)
    with ops.use_kernels(OracleKernels):
        for c in G.manifest()["prefilter"]:
            inp = torch.from_numpy(G.arr(c["inp"]))
            if c["fn"] == "spline_coeff":
                got = interpol.spline_coeff(inp, interpolation=c["order"], bound=c["bound"], dim=c["dim"])
            else:
                got = interpol.spline_coeff_nd(inp, interpolation=c["order"], bound=c["bound"], dim=c["dim"])
            assert G.rel_err(got.numpy(), G.arr(c["out"])) < 1e-11, c
        with pytest.raises(NotImplementedError):
            interpol.spline_coeff_nd(x, interpolation=3, bound='dst2', dim=2)
        with pytest.raises(NotImplementedError):
            interpol.spline_coeff(x, interpolation=2, bound='dst1')
        # orders 0/1 are no-ops that still copy (coeff.py:306-307)
        y = interpol.spline_coeff_nd(x, interpolation=1, bound='dst2', dim=2)
        assert torch.equal(y, x) and y.data_ptr() != x.data_ptr()
        # in-place
        z = x.clone()
        r = interpol.spline_coeff_nd(z, interpolation=3, bound='dct2', dim=2, inplace=True)
        assert r.data_ptr() == z.data_ptr() and not torch.equal(z, x)
        # gradient of the (symmetric) filter = the filter
        xr = x.clone().requires_grad_(True)
        interpol.spline_coeff_nd(xr, 3, 'dct2', 2).backward(torch.ones_like(x))
        want = interpol.spline_coeff_nd(torch.ones_like(x), 3, 'dct2', 2)
        assert torch.allclose(xr.grad, want, atol=1e-12)


@pytest.mark.parametrize("length", [1, 2, 3, 7, 9, 11])
@pytest.mark.parametrize("bound", ["dct1", "dct2", "dft"])
def test_resize_identity_property(length, bound):
    """The reference's tests/test_coeff.py::test_identity, seeded, rtol 1e-4:
    prefilter followed by sampling on the identity lattice returns the input."""
    torch.manual_seed(100 + length)
    x = torch.randn([1, 1, length], dtype=torch.float64)
    with ops.use_kernels(OracleKernels):
        for order in range(8):
            y = interpol.resize(x, shape=[length], bound=bound, interpolation=order)
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-7), (order, bound, length)


def test_resize_restrict_golden_host_logic():
    """`resize` / `restrict` (lattice construction, anchors, defaults, fold/unfold and the
    SeparableGrid they hand to the operators) against the reference's outputs, with the
    kernels served by the oracle."""
    import golden_util as G
    with ops.use_kernels(OracleKernels):
        for c in G.resize_cases():
            fn = getattr(interpol, c["fn"])
            got64 = fn(torch.from_numpy(c["inp"]).double(), **c["kwargs"])
            assert list(got64.shape) == c["shape"], c["kwargs"]
            assert G.rel_err(got64.numpy(), c["out64"]) < 1e-10, (c["fn"], c["kwargs"])
            got32 = fn(torch.from_numpy(c["inp"]), **c["kwargs"])
            assert got32.dtype == torch.float32
            G.assert_close(got32.numpy(), c["out32"], rtol=1e-5, atol_rel=1e-5, what=str((c["fn"], c["kwargs"])))


def test_separable_grid_equals_dense_grid_host_logic():
    from interpol import SeparableGrid
    torch.manual_seed(3)
    x = torch.randn(2, 3, 6, 7, dtype=torch.float64, requires_grad=True)
    lin = [torch.linspace(-1, 6, 9, dtype=torch.float64), torch.linspace(0.5, 5.5, 4, dtype=torch.float64)]
    sep = SeparableGrid(lin)
    assert tuple(sep.shape) == (1, 9, 4, 2) and not sep.requires_grad
    dense = sep.dense()[0]
    with ops.use_kernels(OracleKernels):
        a = interpol.grid_pull(x, sep, interpolation=2, bound='dct2', extrapolate=True)
        b = interpol.grid_pull(x, dense, interpolation=2, bound='dct2', extrapolate=True)
        assert torch.equal(a, b)
        ga, = torch.autograd.grad(a.square().sum(), x)
        gb, = torch.autograd.grad(b.square().sum(), x)
        assert torch.allclose(ga, gb, atol=1e-12)
        y = torch.randn(2, 3, 9, 4, dtype=torch.float64)
        pa = interpol.grid_push(y, sep, shape=[6, 7], interpolation=1, bound='zero')
        pb = interpol.grid_push(y, dense, shape=[6, 7], interpolation=1, bound='zero')
        assert torch.allclose(pa, pb, atol=1e-12)
    with pytest.raises(ValueError):
        SeparableGrid([torch.zeros(2, 2)])


def test_displacement_keyword_host_logic():
    """displacement=True == the same call on add_identity_grid(disp), forward and gradients
    (row f3: the identity lattice is added inside the operator)."""
    torch.manual_seed(5)
    x = torch.randn(2, 2, 7, 8, dtype=torch.float64, requires_grad=True)
    disp = (torch.randn(2, 7, 8, 2, dtype=torch.float64) * 1.5).requires_grad_(True)
    kw = dict(interpolation=3, bound='dct2', extrapolate=True)
    with ops.use_kernels(OracleKernels):
        for fn, args in ((interpol.grid_pull, (x,)), (interpol.grid_grad, (x,)), (interpol.grid_push, (x,)), (interpol.grid_count, ())):
            a = fn(*args, disp, displacement=True, **kw)
            b = fn(*args, interpol.add_identity_grid(disp), **kw)
            assert torch.allclose(a, b, atol=1e-12), fn.__name__
            ins = [t for t in (*args, disp)]
            ga = torch.autograd.grad(a.square().sum(), ins, allow_unused=True)
            gb = torch.autograd.grad(b.square().sum(), ins, allow_unused=True)
            for u, v in zip(ga, gb):
                assert (u is None) == (v is None)
                if u is not None:
                    assert torch.allclose(u, v, atol=1e-10), fn.__name__
    # the reference's positional Function signature is still accepted
    from interpol.autograd import GridPull
    with ops.use_kernels(OracleKernels):
        y = GridPull.apply(x, interpol.add_identity_grid(disp.detach()), 1, 'zero', False)
        y.sum().backward()


def test_label_map_golden_host_logic():
    """Integer inputs: api.grid_pull returns the reference's labels (api.py:194-205) -- through the
    one-pass operator where it applies (orders with <= 27 taps) and the per-label loop elsewhere --
    with the kernels served by the oracle.  Mismatches are allowed only where two labels tie to
    within float rounding (none in these fixtures)."""
    import golden_util as G
    with ops.use_kernels(OracleKernels):
        for c in G.label_cases():
            lab, grid = torch.from_numpy(c["lab"]), torch.from_numpy(c["grid"])
            got = interpol.grid_pull(lab, grid, interpolation=c["order"], bound=c["bound"], extrapolate=c["extrapolate"])
            assert got.dtype == lab.dtype and list(got.shape) == list(c["out"].shape)
            assert np.array_equal(got.numpy(), c["out"]), (c["dim"], c["order"], c["bound"], c["extrapolate"])


def test_grid_helpers():
    g = interpol.identity_grid([3, 4])
    assert g.shape == (3, 4, 2) and g[2, 3].tolist() == [2.0, 3.0]
    d = torch.zeros(2, 3, 4, 2)
    assert torch.equal(interpol.add_identity_grid(d)[1], g)
    assert torch.equal(d, torch.zeros(2, 3, 4, 2))
    interpol.add_identity_grid_(d)
    assert torch.equal(d[0], g)
    mat = torch.tensor([[2., 0., 1.], [0., 1., -1.], [0., 0., 1.]])
    a = interpol.affine_grid(mat, [3, 4])
    assert torch.allclose(a[..., 0], 2 * g[..., 0] + 1) and torch.allclose(a[..., 1], g[..., 1] - 1)
    with pytest.raises(ValueError):
        interpol.affine_grid(mat, [3, 4, 5])


def test_double_backward_flows_through_the_operators():
    """create_graph=True: the backward passes are composed from the differentiable Functions
    (reference pushpull.py:237-325 is plain torch and differentiates twice); third order and beyond through
    grid_grad's backward too (round 5)."""
    import interpol
    torch.manual_seed(3)
    x = torch.randn(1, 2, 5, 6, dtype=torch.float64, requires_grad=True)
    g = (torch.rand(1, 4, 3, 2, dtype=torch.float64) * 3 + 0.7).requires_grad_(True)
    kw = dict(interpolation=3, bound="dct2", extrapolate=True)
    with ops.use_kernels(OracleKernels):
        assert torch.autograd.gradgradcheck(lambda a, b: interpol.grid_pull(a, b, **kw), (x, g), atol=1e-6, rtol=1e-4)
        v = torch.randn(1, 2, 4, 3, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradgradcheck(lambda a, b: interpol.grid_push(a, b, [5, 6], **kw), (v, g), atol=1e-6, rtol=1e-4)
        assert torch.autograd.gradgradcheck(lambda b: interpol.grid_count(b, [5, 6], **kw), (g,), atol=1e-6, rtol=1e-4)
        # a gradient penalty: d/dx |d pull / d grid|^2 exists and matches finite differences
        y = interpol.grid_pull(x, g, **kw)
        gg, = torch.autograd.grad(y.sum(), g, create_graph=True)
        pen = gg.square().sum()
        gx, = torch.autograd.grad(pen, x)
        eps = 1e-6
        d = torch.randn_like(x)

        def penalty(xx):
            gg2, = torch.autograd.grad(interpol.grid_pull(xx, g, **kw).sum(), g, create_graph=True)
            return gg2.square().sum()
        fd = (penalty(x + eps * d) - penalty(x - eps * d)) / (2 * eps)
        fd = float(fd.detach())
        assert abs(fd - float((gx * d).sum())) < 1e-5 * max(1.0, abs(fd))
        # third order through grid_grad's backward (round 5: autograd differentiates the torch restatement of grid_grad, as the
        # reference differentiates its own plain-torch backward, pushpull.py:303-325): d/dg and d/dx of sum(d |grid_grad|^2 / dg)
        # against finite differences, and once more (fourth order) for the shape of it
        def first(xx, gg_):
            z = interpol.grid_grad(xx, gg_, **kw)
            return torch.autograd.grad(z.square().sum(), gg_, create_graph=True)[0]
        g1 = first(x, g)
        h_g, h_x = torch.autograd.grad(g1.sum(), (g, x), create_graph=True)
        dg, dx = torch.randn_like(g), torch.randn_like(x)
        eps = 1e-5
        fd_g = float((first(x, g + eps * dg).sum() - first(x, g - eps * dg).sum()).detach()) / (2 * eps)
        fd_x = float((first(x + eps * dx, g).sum() - first(x - eps * dx, g).sum()).detach()) / (2 * eps)
        assert abs(fd_g - float((h_g * dg).sum().detach())) < 1e-5 * max(1.0, abs(fd_g))
        assert abs(fd_x - float((h_x * dx).sum().detach())) < 1e-5 * max(1.0, abs(fd_x))
        fourth, = torch.autograd.grad(h_g.square().sum(), g)
        assert fourth.shape == g.shape and bool(torch.isfinite(fourth).all())


def test_affine_grid_lazy_lattice_host_logic():
    """AffineGrid quacks like a constant (1, *shape, D) grid: shape conventions, no gradient, dense() equals
    the reference construction affine_grid(mat, shape) when the products are exact."""
    import interpol
    mat = torch.tensor([[0.5, 0.25, 1.5], [-0.125, 1.0, 2.0]], dtype=torch.float64)
    lz = interpol.AffineGrid(mat, [7, 9])
    assert tuple(lz.shape) == (1, 7, 9, 2) and lz.dim() == 4 and not lz.requires_grad
    assert torch.equal(lz.dense()[0], interpol.affine_grid(mat, [7, 9]))
    x = torch.randn(3, 2, 5, 6, dtype=torch.float64, requires_grad=True)
    with ops.use_kernels(OracleKernels):
        y = interpol.grid_pull(x, lz, interpolation=2, bound="dct1", extrapolate=True)
        assert y.shape == (3, 2, 7, 9)
        want = interpol.grid_pull(x, interpol.affine_grid(mat, [7, 9]), interpolation=2, bound="dct1", extrapolate=True)
        assert torch.allclose(y, want, rtol=0, atol=1e-13)
        y.sum().backward()
        assert x.grad is not None
        c = interpol.grid_count(lz, [5, 6], interpolation=1, bound="zero", extrapolate=True)
        assert c.shape == (5, 6)
    with pytest.raises(ValueError):
        interpol.AffineGrid(torch.zeros(2, 3, 4), [3, 3, 3])
