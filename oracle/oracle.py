"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/interpol_oracle.c).

The functions mirror the operator seam of the reference
(`/root/reference/interpol/pushpull.py:35-325`, `coeff.py:288-347`): tensors are
laid out `(B, C, *spatial)` / `(B, *spatial, D)`, bounds/orders are lists of the
integer codes of `bounds.py:8-15` / `splines.py:7-15`, `extrapolate` is 0/1/2.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Cfg(ctypes.Structure):
    _fields_ = [
        ("dim", ctypes.c_int),
        ("bound", ctypes.c_int * 3),
        ("order", ctypes.c_int * 3),
        ("extrapolate", ctypes.c_int),
        ("force_nd", ctypes.c_int),
        ("threads", ctypes.c_int),
        ("vol_bcast", ctypes.c_int),
        ("grid_bcast", ctypes.c_int),
        ("val_bcast", ctypes.c_int),
        ("_pad", ctypes.c_int),
        ("B", ctypes.c_long),
        ("C", ctypes.c_long),
        ("vol_shape", ctypes.c_long * 3),
        ("vol_numel", ctypes.c_long),
        ("n_samples", ctypes.c_long),
    ]


def build(force=False):
    """Compile oracle/liboracle.so with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("interpol_oracle.c", "interpol_oracle_body.inc")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_bound_index.restype = ctypes.c_long
        _LIB.oracle_bound_index.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long]
        _LIB.oracle_bound_sign.restype = ctypes.c_int
        _LIB.oracle_bound_sign.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long]
        for sfx, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            for name in ("weight", "wgrad", "whess"):
                fn = getattr(_LIB, "oracle_%s_%s" % (name, sfx))
                fn.restype = ct
                fn.argtypes = [ctypes.c_int, ct]
    return _LIB


DEFAULT_THREADS = 1


def set_threads(n):
    global DEFAULT_THREADS
    DEFAULT_THREADS = int(n)


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------

def _pad(x, dim):
    """jit_utils.py:9-15 pad_list_int: pad with last, truncate if longer."""
    x = list(x) if isinstance(x, (list, tuple)) else [x]
    x = [int(v) for v in x]
    if len(x) < dim:
        x = x + x[-1:] * (dim - len(x))
    return x[:dim]


class _Torchish:
    """Round-trip torch CPU tensors through numpy without importing torch eagerly."""

    def __init__(self, *xs):
        self.is_torch = any(type(x).__module__.startswith("torch") for x in xs if x is not None)

    def inp(self, x):
        if x is None:
            return None
        if type(x).__module__.startswith("torch"):
            x = x.detach().cpu().numpy()
        return np.asarray(x)

    def out(self, x):
        if self.is_torch:
            import torch
            return torch.from_numpy(x)
        return x


def _sfx(dt):
    if dt == np.float32:
        return "f32"
    if dt == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64 only, got %s" % dt)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _cfg(dim, bound, order, extrapolate, B, C, vol_shape, n_samples,
         vol_bcast=0, grid_bcast=0, val_bcast=0, force_nd=False, threads=None):
    if dim not in (1, 2, 3):
        raise NotImplementedError("oracle: D must be 1, 2 or 3")
    c = _Cfg()
    c.dim = dim
    b = _pad(bound, dim)
    o = _pad(order, dim)
    for d in range(3):
        c.bound[d] = b[d] if d < dim else 1
        c.order[d] = o[d] if d < dim else 0
        c.vol_shape[d] = int(vol_shape[d]) if d < dim else 1
    for k in o:
        if not 0 <= k <= 7:
            raise NotImplementedError("spline order > 7")       # splines.py:80
    c.extrapolate = int(extrapolate)
    c.force_nd = int(bool(force_nd))
    c.threads = int(threads or DEFAULT_THREADS)
    c.vol_bcast, c.grid_bcast, c.val_bcast = int(vol_bcast), int(grid_bcast), int(val_bcast)
    c.B, c.C = int(B), int(C)
    c.vol_numel = int(np.prod([int(s) for s in vol_shape[:dim]]))
    c.n_samples = int(n_samples)
    return c


def _prep_gather(inp, grid):
    dim = grid.shape[-1]
    dt = np.result_type(inp.dtype, grid.dtype)
    inp = np.ascontiguousarray(inp, dtype=dt)
    grid = np.ascontiguousarray(grid, dtype=dt)
    ishape = inp.shape[-dim:]
    oshape = grid.shape[1:-1]
    B = max(inp.shape[0], grid.shape[0])
    C = inp.shape[1]
    return inp, grid, dim, dt, ishape, oshape, B, C


# ----------------------------------------------------------------------------
# operator seam (pushpull.py:35-233)
# ----------------------------------------------------------------------------

def grid_pull(inp, grid, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:35-66 -> nd.py:80 / iso1.pull*d / iso0.pull*d."""
    t = _Torchish(inp, grid)
    inp, grid = t.inp(inp), t.inp(grid)
    inp, grid, dim, dt, ishape, oshape, B, C = _prep_gather(inp, grid)
    N = int(np.prod(oshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, C, ishape, N,
             vol_bcast=(inp.shape[0] == 1 and B > 1), grid_bcast=(grid.shape[0] == 1 and B > 1),
             force_nd=force_nd, threads=threads)
    out = np.empty((B, C) + tuple(oshape), dtype=dt)
    getattr(lib(), "oracle_pull_" + _sfx(dt))(ctypes.byref(c), _ptr(inp), _ptr(grid), _ptr(out))
    return t.out(out)


def grid_grad(inp, grid, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:146-172 -> nd.py:216 / iso1.grad*d / iso0.grad."""
    t = _Torchish(inp, grid)
    inp, grid = t.inp(inp), t.inp(grid)
    inp, grid, dim, dt, ishape, oshape, B, C = _prep_gather(inp, grid)
    N = int(np.prod(oshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, C, ishape, N,
             vol_bcast=(inp.shape[0] == 1 and B > 1), grid_bcast=(grid.shape[0] == 1 and B > 1),
             force_nd=force_nd, threads=threads)
    out = np.empty((B, C) + tuple(oshape) + (dim,), dtype=dt)
    getattr(lib(), "oracle_grad_" + _sfx(dt))(ctypes.byref(c), _ptr(inp), _ptr(grid), _ptr(out))
    return t.out(out)


def grid_hess(inp, grid, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:207-233 -> nd.py:367 / iso1.hess*d / iso0.hess."""
    t = _Torchish(inp, grid)
    inp, grid = t.inp(inp), t.inp(grid)
    inp, grid, dim, dt, ishape, oshape, B, C = _prep_gather(inp, grid)
    N = int(np.prod(oshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, C, ishape, N,
             vol_bcast=(inp.shape[0] == 1 and B > 1), grid_bcast=(grid.shape[0] == 1 and B > 1),
             force_nd=force_nd, threads=threads)
    out = np.empty((B, C) + tuple(oshape) + (dim, dim), dtype=dt)
    getattr(lib(), "oracle_hess_" + _sfx(dt))(ctypes.byref(c), _ptr(inp), _ptr(grid), _ptr(out))
    return t.out(out)


def grid_push(inp, grid, shape, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:70-102 -> nd.py:146 / iso1.push*d / iso0.push*d."""
    t = _Torchish(inp, grid)
    inp, grid = t.inp(inp), t.inp(grid)
    dim = grid.shape[-1]
    dt = np.result_type(inp.dtype, grid.dtype)
    inp = np.ascontiguousarray(inp, dtype=dt)
    grid = np.ascontiguousarray(grid, dtype=dt)
    gshape = grid.shape[1:-1]
    if tuple(inp.shape[-dim:]) != tuple(gshape):
        raise ValueError("Input and grid should have the same spatial shape")  # iso1.py:149-150
    if shape is None:
        shape = gshape
    shape = [int(s) for s in shape]
    B = max(inp.shape[0], grid.shape[0])
    C = inp.shape[1]
    N = int(np.prod(gshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, C, shape, N,
             val_bcast=(inp.shape[0] == 1 and B > 1), grid_bcast=(grid.shape[0] == 1 and B > 1),
             force_nd=force_nd, threads=threads)
    out = np.empty((B, C) + tuple(shape), dtype=dt)
    getattr(lib(), "oracle_push_" + _sfx(dt))(ctypes.byref(c), _ptr(inp), _ptr(grid), _ptr(out))
    return t.out(out)


def grid_count(grid, shape, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:106-142: push of an all-ones (B,1,*spatial) image."""
    t = _Torchish(grid)
    grid = np.ascontiguousarray(t.inp(grid))
    dim = grid.shape[-1]
    dt = grid.dtype
    gshape = grid.shape[1:-1]
    if shape is None:
        shape = gshape
    shape = [int(s) for s in shape]
    B = grid.shape[0]
    N = int(np.prod(gshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, 1, shape, N,
             force_nd=force_nd, threads=threads)
    out = np.empty((B, 1) + tuple(shape), dtype=dt)
    getattr(lib(), "oracle_push_" + _sfx(dt))(ctypes.byref(c), None, _ptr(grid), _ptr(out))
    return t.out(out)


def grid_pushgrad(inp, grid, shape, bound, interpolation, extrapolate, force_nd=False, threads=None):
    """pushpull.py:176-203 -> nd.py:291 / iso1.pushgrad*d / iso0.pushgrad."""
    t = _Torchish(inp, grid)
    inp, grid = t.inp(inp), t.inp(grid)
    dim = grid.shape[-1]
    dt = np.result_type(inp.dtype, grid.dtype)
    inp = np.ascontiguousarray(inp, dtype=dt)
    grid = np.ascontiguousarray(grid, dtype=dt)
    gshape = grid.shape[1:-1]
    if shape is None:
        shape = gshape
    shape = [int(s) for s in shape]
    B = max(inp.shape[0], grid.shape[0])
    C = inp.shape[1]
    N = int(np.prod(gshape))
    c = _cfg(dim, bound, interpolation, extrapolate, B, C, shape, N,
             val_bcast=(inp.shape[0] == 1 and B > 1), grid_bcast=(grid.shape[0] == 1 and B > 1),
             force_nd=force_nd, threads=threads)
    out = np.empty((B, C) + tuple(shape), dtype=dt)
    getattr(lib(), "oracle_pushgrad_" + _sfx(dt))(ctypes.byref(c), _ptr(inp), _ptr(grid), _ptr(out))
    return t.out(out)


# ----------------------------------------------------------------------------
# backward compositions (pushpull.py:237-325)
# ----------------------------------------------------------------------------

def _np(x):
    return x.detach().cpu().numpy() if type(x).__module__.startswith("torch") else np.asarray(x)


def grid_pull_backward(grad, inp, grid, bound, interpolation, extrapolate, **kw):
    """pushpull.py:237-258: returns (grad_inp, grad_grid)."""
    t = _Torchish(grad, inp, grid)
    dim = grid.shape[-1]
    g_inp = grid_push(_np(grad), _np(grid), list(inp.shape[-dim:]), bound, interpolation, extrapolate, **kw)
    gg = grid_grad(_np(inp), _np(grid), bound, interpolation, extrapolate, **kw)
    g_grid = (gg * _np(grad)[..., None]).sum(axis=1)
    return t.out(np.ascontiguousarray(g_inp)), t.out(np.ascontiguousarray(g_grid))


def grid_push_backward(grad, inp, grid, bound, interpolation, extrapolate, **kw):
    """pushpull.py:262-282."""
    t = _Torchish(grad, inp, grid)
    g_inp = grid_pull(_np(grad), _np(grid), bound, interpolation, extrapolate, **kw)
    gg = grid_grad(_np(grad), _np(grid), bound, interpolation, extrapolate, **kw)
    g_grid = (gg * _np(inp)[..., None]).sum(axis=1)
    return t.out(np.ascontiguousarray(g_inp)), t.out(np.ascontiguousarray(g_grid))


def grid_count_backward(grad, grid, bound, interpolation, extrapolate, **kw):
    """pushpull.py:286-299."""
    t = _Torchish(grad, grid)
    gg = grid_grad(_np(grad), _np(grid), bound, interpolation, extrapolate, **kw)
    return t.out(np.ascontiguousarray(gg.sum(axis=1)))


def grid_grad_backward(grad, inp, grid, bound, interpolation, extrapolate, **kw):
    """pushpull.py:303-325."""
    t = _Torchish(grad, inp, grid)
    dim = grid.shape[-1]
    g_inp = grid_pushgrad(_np(grad), _np(grid), list(inp.shape[-dim:]), bound, interpolation, extrapolate, **kw)
    hh = grid_hess(_np(inp), _np(grid), bound, interpolation, extrapolate, **kw)
    g_grid = (hh * _np(grad)[..., None]).sum(axis=(1, -2))
    return t.out(np.ascontiguousarray(g_inp)), t.out(np.ascontiguousarray(g_grid))


# ----------------------------------------------------------------------------
# prefilter (coeff.py:288-347)
# ----------------------------------------------------------------------------

def spline_coeff(inp, bound, order, dim=-1, threads=None):
    """coeff.py:288-313, out of place."""
    t = _Torchish(inp)
    a = np.array(t.inp(inp), copy=True, order="C")
    if isinstance(bound, (list, tuple)):
        bound = bound[0]
    if isinstance(order, (list, tuple)):
        order = order[0]
    if order > 7:
        raise NotImplementedError
    if order in (0, 1):
        return t.out(a)
    dim = dim % a.ndim
    n = a.shape[dim]
    outer = int(np.prod(a.shape[:dim])) if dim > 0 else 1
    inner = int(np.prod(a.shape[dim + 1:])) if dim < a.ndim - 1 else 1
    rc = getattr(lib(), "oracle_spline_filter_" + _sfx(a.dtype))(
        _ptr(a), ctypes.c_long(outer), ctypes.c_long(n), ctypes.c_long(inner),
        int(bound), int(order), int(threads or DEFAULT_THREADS))
    if rc != 0:
        raise NotImplementedError("prefilter bound %d" % bound)     # coeff.py:243-244
    return t.out(a)


def spline_coeff_nd(inp, bound, order, dim=None, threads=None):
    """coeff.py:317-347: filter the last `dim` dims (all dims when None)."""
    t = _Torchish(inp)
    a = t.inp(inp)
    if dim is None:
        dim = a.ndim
    b = _pad(bound, dim)
    o = _pad(order, dim)
    for d in range(dim):
        a = spline_coeff(a, b[d], o[d], dim=-dim + d, threads=threads)
    return t.out(np.array(a, copy=True) if dim == 0 else a)


# ----------------------------------------------------------------------------
# scalar primitives (bounds.py / splines.py)
# ----------------------------------------------------------------------------

def bound_index(bound, i, n):
    return int(lib().oracle_bound_index(int(bound), int(i), int(n)))


def bound_sign(bound, i, n):
    """Returns None when Bound.transform returns None."""
    s = int(lib().oracle_bound_sign(int(bound), int(i), int(n)))
    return None if s == 2 else s


def weight(order, x, dtype="f64"):
    return float(getattr(lib(), "oracle_weight_" + dtype)(int(order), x))


def wgrad(order, x, dtype="f64"):
    return float(getattr(lib(), "oracle_wgrad_" + dtype)(int(order), x))


def whess(order, x, dtype="f64"):
    return float(getattr(lib(), "oracle_whess_" + dtype)(int(order), x))
