"""Pins the CPU oracle against the committed golden vectors (runs anywhere)."""
import numpy as np
import pytest

import golden_util as G
from oracle import oracle


def _cases(op):
    return [c for c in G.manifest()["cases"] if c["op"] == op]


@pytest.mark.parametrize("op", ["pull", "push", "count", "grad", "pushgrad", "hess"])
def test_ops_fp64(op):
    cases = _cases(op)
    assert len(cases) > 50
    for c in cases:
        got = G.run_case(oracle, c, np.float64)
        assert G.rel_err(got, G.arr(c["output"])) < 1e-12, c


@pytest.mark.parametrize("op", ["pull", "push", "count", "grad", "pushgrad", "hess"])
def test_ops_fp32(op):
    for c in _cases(op):
        got = G.run_case(oracle, c, np.float32)
        assert got.dtype == np.float32
        rtol, atol_rel = G.reference_fp32_tol(c)           # (the oracle's fp32 mode IS the reference's fp32 arithmetic)
        G.assert_close(got, G.arr(c["output"]), rtol=rtol, atol_rel=atol_rel, what=str(c))


def test_nearest_pull_bit_exact_fp32():
    """Order-0 pull only copies (and sign-flips) values: must be bit-exact."""
    n = 0
    for c in _cases("pull"):
        if all(o == 0 for o in c["order"][:c["dim"]]):
            got = G.run_case(oracle, c, np.float32)
            want = G.arr(c["output"]).astype(np.float32)
            assert np.array_equal(got, want), c
            n += 1
    assert n >= 20


def test_prefilter():
    for c in G.manifest()["prefilter"]:
        x = G.arr(c["inp"])
        if c["fn"] == "spline_coeff":
            got = oracle.spline_coeff(x, c["bound"], c["order"], dim=c["dim"])
        else:
            got = oracle.spline_coeff_nd(x, c["bound"], c["order"], c["dim"])
        assert G.rel_err(got, G.arr(c["out"])) < 1e-11, c


def test_backward_compositions():
    """API-level autograd results of the reference == oracle compositions."""
    from interpol_codes import to_int_lists
    for c in G.manifest()["backward"]:
        b, o = to_int_lists(c["bound"], c["interpolation"])
        gout = G.arr(c["gout"])
        grid = G.arr(c["grid"])
        if c["fn"] == "grid_pull":
            inp = G.arr(c["inp"])
            gi, gg = oracle.grid_pull_backward(gout, inp, grid, b, o, 1)
        elif c["fn"] == "grid_push":
            inp = G.arr(c["inp"])
            gi, gg = oracle.grid_push_backward(gout, inp, grid, b, o, 1)
        elif c["fn"] == "grid_count":
            gi, gg = None, oracle.grid_count_backward(gout, grid, b, o, 1)
        else:
            inp = G.arr(c["inp"])
            gi, gg = oracle.grid_grad_backward(gout, inp, grid, b, o, 1)
        if gi is not None:
            assert G.rel_err(gi, G.arr(c["grad_inp"])) < 1e-12, c
        assert G.rel_err(gg, G.arr(c["grad_grid"])) < 1e-12, c


def test_index_sign_tables():
    for key, t in G.api()["tables"].items():
        n, b = int(key[1:key.index("_")]), int(key[-1])
        for k, want in enumerate(t["idx"]):
            i = t["i0"] + k
            assert oracle.bound_index(b, i, n) == want
            s = oracle.bound_sign(b, i, n)
            assert s == (None if t["sign"] is None else t["sign"][k])


def test_kat_1d():
    api = G.api()
    x = np.array([[[1., 2., 3., 4.]]])
    coords = np.array(api["kat_coords"], dtype=np.float64).reshape(1, -1, 1)
    for b in range(7):
        for o in (0, 1):
            got = oracle.grid_pull(x, coords, [b], [o], 1).reshape(-1)
            assert np.allclose(got, api["kat"]["b%d_o%d" % (b, o)], atol=1e-14)
    c2 = np.array(api["kat_extrap_coords"]).reshape(1, -1, 1)
    for ex in (0, 1, 2):
        got = oracle.grid_pull(x, c2, [1], [1], ex).reshape(-1)
        assert np.allclose(got, api["kat"]["extrap%d" % ex], atol=1e-14)


def test_mid_size_cases():
    """The oracle against the mid-size reference vectors (the ones that reach the tile kernels on the GPU)."""
    man, npz = G.mid()
    assert len(man["cases"]) >= 80
    for c in man["cases"]:
        ins = {k: np.asarray(npz[v], dtype=np.float64) for k, v in c["inputs"].items()}
        b, o, e = c["bound"], c["order"], c["extrapolate"]
        if c["op"] == "pull":
            got = oracle.grid_pull(ins["inp"], ins["grid"], b, o, e)
        elif c["op"] == "grad":
            got = oracle.grid_grad(ins["inp"], ins["grid"], b, o, e)
        elif c["op"] == "push":
            got = oracle.grid_push(ins["inp"], ins["grid"], c["shape"], b, o, e)
        else:
            got = oracle.grid_count(ins["grid"], c["shape"], b, o, e)
        # expected values were stored as float32: 6e-8 relative
        G.assert_close(got, npz[c["output"]], rtol=2e-7, atol_rel=2e-7, what=str(c))


def test_fold_cases():
    """The oracle against the reference vectors on lattices of 33 - 40 points with overhanging sample grids (the cases
    that meet the folding end bricks of the owner-computes push on the GPU)."""
    man, npz = G.fold()
    assert len(man["cases"]) >= 15
    for c in man["cases"]:
        ins = {k: np.asarray(npz[v], dtype=np.float64) for k, v in c["inputs"].items()}
        b, o, e = c["bound"], c["order"], c["extrapolate"]
        got = oracle.grid_push(ins["inp"], ins["grid"], c["shape"], b, o, e) if c["op"] == "push" else oracle.grid_count(ins["grid"], c["shape"], b, o, e)
        G.assert_close(got, npz[c["output"]], rtol=2e-7, atol_rel=2e-7, what=str(c))


def test_mid_size_backward():
    from interpol_codes import to_int_lists
    man, npz = G.mid()
    for c in man["backward"]:
        b, o = to_int_lists(c["bound"], c["interpolation"])
        f = lambda k: np.asarray(npz[c[k]], dtype=np.float64)
        if c["fn"] == "grid_pull":
            gi, gg = oracle.grid_pull_backward(f("gout"), f("inp"), f("grid"), b, o, 1)
        else:
            gi, gg = oracle.grid_push_backward(f("gout"), f("inp"), f("grid"), b, o, 1)
        G.assert_close(gi, npz[c["grad_inp"]], rtol=2e-7, atol_rel=2e-7, what=c["fn"] + " grad_inp")
        G.assert_close(gg, npz[c["grad_grid"]], rtol=2e-7, atol_rel=2e-7, what=c["fn"] + " grad_grid")
