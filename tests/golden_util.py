"""Loader for the golden vectors of tests/golden/ (generated from the reference
by tests/golden/make_golden.py in the build container)."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CACHE = {}


def arrays():
    if "npz" not in _CACHE:
        _CACHE["npz"] = np.load(os.path.join(HERE, "golden_ops.npz"))
    return _CACHE["npz"]


def manifest():
    if "ops" not in _CACHE:
        with open(os.path.join(HERE, "golden_ops.json")) as f:
            _CACHE["ops"] = json.load(f)
    return _CACHE["ops"]


def api():
    if "api" not in _CACHE:
        with open(os.path.join(HERE, "golden_api.json")) as f:
            _CACHE["api"] = json.load(f)
    return _CACHE["api"]


def mid():
    """Mid-size cases (>= 4096 samples: they reach the LDS-tile kernels), tests/golden/make_golden_mid.py.
    Returns (manifest, arrays); inputs are float32, expected outputs the reference's float64 results
    stored as float32."""
    if "mid" not in _CACHE:
        with open(os.path.join(HERE, "golden_mid.json")) as f:
            _CACHE["mid"] = (json.load(f), np.load(os.path.join(HERE, "golden_mid.npz")))
    return _CACHE["mid"]


def fold():
    """Scatter cases on lattices of 33 - 40 points (the end bricks of the owner-computes push fold), sample grids that
    overhang the lattice, tests/golden/make_golden_fold.py.  Same conventions as mid()."""
    if "fold" not in _CACHE:
        with open(os.path.join(HERE, "golden_fold.json")) as f:
            _CACHE["fold"] = (json.load(f), np.load(os.path.join(HERE, "golden_fold.npz")))
    return _CACHE["fold"]


def arr(name, dtype=np.float64):
    return np.asarray(arrays()[name], dtype=dtype)


def rel_err(got, want):
    """max |got-want| / max|want| (the `atol = tol*max|ref|` criterion of SURVEY A.8)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0
    scale = max(float(np.abs(want).max()), 1e-30)
    return float(np.abs(got - want).max()) / scale


def assert_close(got, want, rtol, atol_rel, what=""):
    """|got - want| <= rtol*|want| + atol_rel*max|want| elementwise."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return
    atol = atol_rel * max(float(np.abs(want).max()), 1e-30)
    bad = np.abs(got - want) > (rtol * np.abs(want) + atol)
    assert not bad.any(), "%s: %d/%d mismatches, max abs err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.size, float(np.abs(got - want).max()), float(np.abs(want).max()))



def f32_masked_samples(grid, shape, extrapolate):
    """Samples that the reference masks when it runs in float32 and a float64 evaluation of the SAME float32 coordinates does
    not (nd.py:10-27): the Python scalar threshold (-thr, n - 1 + thr; thr = 0.05 / 0.55) is cast to the tensor's
    dtype for the comparison, so a float32 coordinate EQUAL to the float32-rounded threshold is outside in float32 and -- where
    the rounding went towards the inside -- still inside in float64.  Returns a bool array (B, *out): expected outputs of
    gathers at those samples are zeros (the mask multiplies the result).  The opposite disagreement cannot occur (no float32
    lies strictly between a double and its nearest float32); asserted."""
    g32 = np.asarray(grid, dtype=np.float32)
    out = np.zeros(g32.shape[:-1], dtype=bool)
    if extrapolate == 1:
        return out
    thr = 0.05 if extrapolate == 0 else 0.55
    m32 = np.ones(g32.shape[:-1], dtype=bool)
    m64 = np.ones(g32.shape[:-1], dtype=bool)
    for d, n in enumerate(shape):
        x32, x64 = g32[..., d], g32[..., d].astype(np.float64)
        m32 &= (x32 > np.float32(-thr)) & (x32 < np.float32(n - 1 + thr))
        m64 &= (x64 > -thr) & (x64 < n - 1 + thr)
    assert not (m32 & ~m64).any()
    return m64 & ~m32

def run_case(ops, case, dtype):
    """Run one manifest case through an operator table `ops` exposing the
    pushpull-style functions grid_pull/grid_push/... on numpy arrays."""
    ins = {k: arr(v, dtype) for k, v in case["inputs"].items()}
    b, o, e = case["bound"], case["order"], case["extrapolate"]
    op = case["op"]
    if op == "pull":
        return ops.grid_pull(ins["inp"], ins["grid"], b, o, e)
    if op == "grad":
        return ops.grid_grad(ins["inp"], ins["grid"], b, o, e)
    if op == "hess":
        return ops.grid_hess(ins["inp"], ins["grid"], b, o, e)
    if op == "push":
        return ops.grid_push(ins["inp"], ins["grid"], case["shape"], b, o, e)
    if op == "count":
        return ops.grid_count(ins["grid"], case["shape"], b, o, e)
    if op == "pushgrad":
        return ops.grid_pushgrad(ins["inp"], ins["grid"], case["shape"], b, o, e)
    raise ValueError(op)


def fp32_tol(case_or_order=None):
    """Stated fp32 parity tolerance (rtol, atol relative to max|ref|): north_star's rtol = 1e-5 with atol = 1e-5 max|ref|
    (SURVEY A.8), for EVERY spline order.  The goldens are the reference's float64 outputs; the kernels evaluate the
    middle pieces of the order 4-7 splines about their outer breakpoints (csrc/spline_math.hpp), which keeps the fp32
    weights within 1e-7 of the exact ones where the reference's own fp32 Horner forms (splines.py:56-79) lose five
    digits.  Measured worst errors, in units of this tolerance: tests/tolerance_report.py -> profiles/r04_tolerance.txt."""
    return (1e-5, 1e-5)


def reference_fp32_tol(case_or_order):
    """What the REFERENCE's own fp32 arithmetic achieves against its fp64 outputs (the oracle's fp32 mode restates it,
    tests/test_oracle_golden.py): rtol 1e-5 with atol 1e-5 max|ref| up to order 5; for orders 6 and 7 its Horner polynomials
    (splines.py:56-79) cancel near the knots and it deviates by up to 1.3e-5 max|ref| on these vectors: 5e-5 there.  The HIP
    kernels are held to fp32_tol() for every order."""
    order = case_or_order["order"] if isinstance(case_or_order, dict) else case_or_order
    order = order if isinstance(order, (list, tuple)) else [order]
    return (1e-5, 5e-5) if max(order) >= 6 else (1e-5, 1e-5)


def resize_cases():
    """resize / restrict golden cases (tests/golden/make_golden_resize.py): list of dicts with
    fn, kwargs, inp (float32), out64 (reference on the float64 input), out32 (on the float32 input)."""
    if "resize" not in _CACHE:
        with open(os.path.join(HERE, "golden_resize.json")) as f:
            man = json.load(f)
        npz = np.load(os.path.join(HERE, "golden_resize.npz"))
        _CACHE["resize"] = [dict(c, inp=npz[c["inp"]], out64=npz[c["out64"]], out32=npz[c["out32"]]) for c in man["cases"]]
    return _CACHE["resize"]


def label_cases():
    """label-map golden cases (tests/golden/make_golden_labels.py): dicts with dim, order, bound,
    extrapolate, lab (int64), grid (float32), out (int64: the reference's result)."""
    if "labels" not in _CACHE:
        with open(os.path.join(HERE, "golden_labels.json")) as f:
            man = json.load(f)
        npz = np.load(os.path.join(HERE, "golden_labels.npz"))
        _CACHE["labels"] = [dict(c, lab=npz["l%d/lab" % c["i"]].astype(np.int64), grid=npz["l%d/grid" % c["i"]],
                                 out=npz["l%d/out" % c["i"]].astype(np.int64)) for c in man["cases"]]
    return _CACHE["labels"]
