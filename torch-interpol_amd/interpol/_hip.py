"""ctypes binding of libinterpol_hip.so (C-ABI: include/interpol_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; the
sampling itself runs in the hand-written gfx950 kernels of `csrc/`.  There is
NO CPU fallback: without the built library, or with CPU tensors, every
operator raises.
"""
import ctypes
import threading
import os

import torch

from .sepgrid import SeparableGrid, AffineGrid, LazyGrid

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# INTERPOL_HIP_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("INTERPOL_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libinterpol_hip.so")

ABI_VERSION = 1
F32, F64, BF16, F16 = 0, 1, 2, 3
FLAG_NO_FASTPATH = 1
FLAG_ACCUMULATE = 2
FLAG_FORCE_TILED = 4
FLAG_SEPARABLE_GRID = 8
FLAG_DISPLACEMENT = 16
FLAG_WITH_COUNT = 32
FLAG_BINNED_SCATTER = 64
FLAG_AUTO_SCATTER = 1 << 24
_ROUTED_2D = (torch.float32, torch.bfloat16, torch.float16)   # storage types of the 2-D bricks (csrc/scatter2d.hip)
FLAG_SMALL_TILES = 1 << 25           # (experimental, opt-in: experiments/pull_direct.hip)
_POISON_SCRATCH = os.environ.get("INTERPOL_POISON_SCRATCH", "0") not in ("", "0")
_WS_NOCACHE = os.environ.get("INTERPOL_WS_NOCACHE", "0") not in ("", "0")     # (debugging aid: a fresh workspace per call, as inside a hipGraph capture)
FLAG_AFFINE_GRID = 128

_DTYPE_CODE = {torch.float32: F32, torch.float64: F64, torch.bfloat16: BF16, torch.float16: F16}

# exported symbols, as declared in include/interpol_hip.h
SYMBOLS = (
    "interpol_pull", "interpol_push", "interpol_count", "interpol_grad", "interpol_pushgrad",
    "interpol_hess", "interpol_pull_backward", "interpol_push_backward", "interpol_count_backward",
    "interpol_spline_filter", "interpol_spline_filter_to", "interpol_resample_1d", "interpol_resample_1d_gathers", "interpol_pull_labels",
    "interpol_push_bricks", "interpol_push_bricks_workspace", "interpol_host_bound_index", "interpol_host_bound_sign",
    "interpol_host_weight", "interpol_host_weight_f32", "interpol_abi_version",
    "interpol_error_string", "interpol_kernel_name", "interpol_scatter_workspace",
    "interpol_set_handback", "interpol_release_stream", "interpol_has_experiments", "interpol_pull_workspace", "interpol_pull_ws", "interpol_push_backward_ws", "interpol_grad_ws",
)


class Problem(ctypes.Structure):
    """`interpol_problem` of include/interpol_hip.h."""
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("dim", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("grid_dtype", ctypes.c_int32),
        ("extrapolate", ctypes.c_int32),
        ("bound", ctypes.c_int32 * 3),
        ("order", ctypes.c_int32 * 3),
        ("flags", ctypes.c_int32),
        ("batch", ctypes.c_int64),
        ("channels", ctypes.c_int64),
        ("vol_shape", ctypes.c_int64 * 3),
        ("grid_shape", ctypes.c_int64 * 3),
        ("vol_stride", ctypes.c_int64 * 5),
        ("grid_stride", ctypes.c_int64 * 5),
        ("val_stride", ctypes.c_int64 * 7),
    ]


_lib = None


class HipExtensionMissing(ImportError):
    pass


# Workspaces of the probe-routed organisations (bricks of the target / of the image) are OPTIONAL: every operator has an
# organisation that needs none.  ONE buffer per (device, stream, HOST THREAD) is kept between calls and grown on demand (pull, push
# and the backward passes of a stream share it: their kernels are ordered by the stream, and a captured hipGraph keeps pointing at
# live memory).  The host thread is part of the key because ctypes releases the GIL inside the library: two threads issuing routed
# operators on ONE stream could otherwise interleave their kernel enqueues on one buffer (A's bin, B's header zeroing and bin, A's
# accumulate).  At most _WS_MAX buffers are kept (least recently used first out -- a warm-up on a side stream, the usual hipGraph
# recipe, does not pin a second 1 - 2 GB buffer for good); `release_workspaces()` gives them all back (call it next to
# `torch.cuda.empty_cache()`).  A miss asks torch's caching allocator, but only when the request fits
# comfortably: asking for more than it can give makes the allocator synchronise the device and flush its cache before it
# raises -- on every call, if the caller runs near capacity.  So a new buffer never takes more than half of what is available
# (free device memory + the allocator's own free blocks) and a request that failed is not repeated until noticeably more
# memory is available.  (The allocator's statistics cost ~0.25 ms of host time: they are consulted on misses only.)
_WS_CACHE = {}                                       # (device index, stream handle, host thread) -> uint8 tensor; insertion order = age
_WS_MAX = 4
_WS_DENIED = {}                                      # device index -> (bytes asked for, bytes available at the time)
_WS_CHECK_ABOVE = 64 << 20


def release_workspaces():
    """Drop the cached workspaces (they return to torch's caching allocator)."""
    _WS_CACHE.clear()
    _WS_DENIED.clear()


def _available(idx):
    free, _ = torch.cuda.mem_get_info(idx)
    return free + torch.cuda.memory_reserved(idx) - torch.cuda.memory_allocated(idx)


def _optional_workspace(nbytes, dev):
    """nbytes of device scratch, or None when the call should do without (no exception, no allocator flush)."""
    if nbytes <= 0:
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing() or _WS_NOCACHE:
        # inside a hipGraph capture the buffer must belong to the graph (its private pool keeps it alive for the replays): a cached
        # buffer could be replaced by a larger one -- and freed -- after the capture
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            return None
    key = (idx, int(torch.cuda.current_stream(dev).cuda_stream), threading.get_ident())
    ws = _WS_CACHE.get(key)
    if ws is not None and ws.numel() >= nbytes:
        _WS_CACHE[key] = _WS_CACHE.pop(key)          # (most recently used: last)
        return ws
    avail = None
    if nbytes > _WS_CHECK_ABOVE:
        if ws is not None:
            del _WS_CACHE[key]                       # (the old buffer goes back to the allocator first: it counts as available)
            ws = None
        avail = _available(idx)
        denied = _WS_DENIED.get(idx)
        if denied is not None and nbytes >= denied[0] and avail < denied[1] + denied[0] // 2:
            return None
        if 2 * nbytes > avail:
            return None
    try:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    except torch.cuda.OutOfMemoryError:
        _WS_DENIED[idx] = (nbytes, avail if avail is not None else _available(idx))
        return None
    if _WS_DENIED:
        _WS_DENIED.pop(idx, None)
    _WS_CACHE.pop(key, None)
    _WS_CACHE[key] = ws
    while len(_WS_CACHE) > _WS_MAX:                  # (the kernels that use an evicted buffer are enqueued: the caching allocator
        _WS_CACHE.pop(next(iter(_WS_CACHE)))         #  hands its memory to later work of the same stream only)
    return ws


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "libinterpol_hip.so not found at %s: build it with `make -C %s -j8` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % (LIB_PATH, _PKG_ROOT))
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    pp = ctypes.POINTER(Problem)
    L.interpol_pull.argtypes = [pp, vp, vp, vp, vp]
    L.interpol_grad.argtypes = [pp, vp, vp, vp, vp]
    L.interpol_hess.argtypes = [pp, vp, vp, vp, vp]
    L.interpol_push.argtypes = [pp, vp, vp, vp, vp, i64, vp]
    L.interpol_pushgrad.argtypes = [pp, vp, vp, vp, vp, i64, vp]
    L.interpol_count.argtypes = [pp, vp, vp, vp, i64, vp]
    L.interpol_pull_backward.argtypes = [pp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.interpol_push_backward.argtypes = [pp, vp, vp, vp, vp, vp, vp]
    L.interpol_count_backward.argtypes = [pp, vp, vp, vp, vp]
    L.interpol_spline_filter.argtypes = [vp, i32, i64, i64, i64, i32, i32, vp]
    L.interpol_spline_filter_to.argtypes = [vp, vp, i32, i64, i64, i64, i32, i32, vp]
    L.interpol_pull_labels.argtypes = [pp, vp, vp, vp, vp]
    L.interpol_push_bricks.argtypes = [pp, vp, vp, vp, vp, i64, vp]
    L.interpol_push_bricks.restype = ctypes.c_int
    L.interpol_push_bricks_workspace.argtypes = [pp]
    L.interpol_push_bricks_workspace.restype = i64
    L.interpol_scatter_workspace.argtypes = [pp, i32]
    L.interpol_scatter_workspace.restype = i64
    L.interpol_pull_workspace.argtypes = [pp]
    L.interpol_pull_workspace.restype = i64
    L.interpol_pull_ws.argtypes = [pp, vp, vp, vp, vp, i64, vp]
    L.interpol_pull_ws.restype = ctypes.c_int
    L.interpol_grad_ws.argtypes = [pp, vp, vp, vp, vp, i64, vp]
    L.interpol_grad_ws.restype = ctypes.c_int
    L.interpol_push_backward_ws.argtypes = [pp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.interpol_push_backward_ws.restype = ctypes.c_int
    L.interpol_set_handback.argtypes = [i32]
    L.interpol_set_handback.restype = i32
    L.interpol_release_stream.argtypes = [ctypes.c_void_p]
    L.interpol_release_stream.restype = i32
    L.interpol_resample_1d.argtypes = [i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, vp, vp, vp, vp]
    L.interpol_resample_1d_gathers.argtypes = [i32, i64, i64]
    L.interpol_resample_1d_gathers.restype = i32
    for name in ("interpol_pull", "interpol_grad", "interpol_hess", "interpol_push", "interpol_pushgrad",
                 "interpol_count", "interpol_pull_backward", "interpol_push_backward",
                 "interpol_count_backward", "interpol_spline_filter", "interpol_spline_filter_to", "interpol_resample_1d", "interpol_pull_labels"):
        getattr(L, name).restype = ctypes.c_int
    L.interpol_host_bound_index.argtypes = [i32, i32, i32]
    L.interpol_host_bound_index.restype = i32
    L.interpol_host_bound_sign.argtypes = [i32, i32, i32]
    L.interpol_host_bound_sign.restype = i32
    L.interpol_host_weight.argtypes = [i32, ctypes.c_double, i32]
    L.interpol_host_weight.restype = ctypes.c_double
    L.interpol_host_weight_f32.argtypes = [i32, ctypes.c_float, i32]
    L.interpol_host_weight_f32.restype = ctypes.c_float
    L.interpol_abi_version.restype = i32
    L.interpol_error_string.argtypes = [ctypes.c_int]
    L.interpol_error_string.restype = ctypes.c_char_p
    L.interpol_kernel_name.argtypes = [pp, ctypes.c_char_p]
    L.interpol_kernel_name.restype = ctypes.c_char_p
    if L.interpol_abi_version() != ABI_VERSION:
        raise HipExtensionMissing("libinterpol_hip.so has ABI version %d, expected %d: rebuild it"
                                  % (L.interpol_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _check(rc, what):
    if rc == 0:
        return
    msg = lib().interpol_error_string(rc).decode()
    if rc == -2:
        raise NotImplementedError("%s: %s" % (what, msg))          # reference interpol/splines.py:80
    if rc == -8:
        raise NotImplementedError("%s: %s" % (what, msg))          # reference interpol/coeff.py:243-244
    if rc in (-1, -3, -5, -7):
        raise ValueError("%s: %s" % (what, msg))
    raise RuntimeError("%s failed: %s (code %d)" % (what, msg, rc))


def _require_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "interpol (MI355X build): tensors must live on a ROCm GPU; got a %s tensor. "
                "This build has no CPU path." % t.device.type)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("interpol: all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def common_dtypes(img, grid):
    """Storage / coordinate dtypes the kernels run in.
    (f32, f32), (f64, f64), (bf16|f16 image, f32 coordinates); anything else is
    promoted the way the reference's elementwise ops would."""
    if not grid.dtype.is_floating_point:
        raise TypeError("grid must be a floating point tensor")
    idt = img.dtype if img is not None else grid.dtype
    if not idt.is_floating_point:
        raise TypeError("image must be a floating point tensor")
    if idt == torch.float64 or grid.dtype == torch.float64:
        return torch.float64, torch.float64
    if idt in (torch.bfloat16, torch.float16):
        return idt, torch.float32
    return torch.float32, torch.float32


def _spatially_contiguous(t, first_spatial):
    """True when dims [first_spatial:] are row-major contiguous."""
    expect = 1
    for d in range(t.dim() - 1, first_spatial - 1, -1):
        if t.shape[d] > 1 and t.stride(d) != expect:
            return False
        expect *= t.shape[d]
    return True


def _bstride(t, B):
    """Batch stride with broadcasting of a singleton batch."""
    return 0 if (t.shape[0] == 1 and B > 1) else t.stride(0)


def make_problem(dim, dtype, grid_dtype, bound, order, extrapolate, B, C, vol_shape, grid_shape,
                 vol_stride, grid_stride, val_stride, flags=0):
    p = Problem()
    p.abi_version = ABI_VERSION
    p.dim = dim
    p.dtype = _DTYPE_CODE[dtype]
    p.grid_dtype = _DTYPE_CODE[grid_dtype]
    p.extrapolate = int(extrapolate)
    for d in range(3):
        p.bound[d] = int(bound[d]) if d < dim else 1
        p.order[d] = int(order[d]) if d < dim else 0
        p.vol_shape[d] = int(vol_shape[d]) if d < dim else 1
        p.grid_shape[d] = int(grid_shape[d]) if d < dim else 1
    p.flags = flags
    p.batch, p.channels = int(B), int(C)
    for i, s in enumerate(vol_stride):
        p.vol_stride[i] = int(s)
    for i, s in enumerate(grid_stride):
        p.grid_stride[i] = int(s)
    for i, s in enumerate(val_stride):
        p.val_stride[i] = int(s)
    return p


def _dense_strides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= int(s)
    return list(reversed(st))


def _pad_to(lst, n, fill=0):
    return list(lst) + [fill] * (n - len(lst))


def _grid_strides(grid, B, dim):
    if isinstance(grid, _PackedLattice):
        return [0] * 5
    return [_bstride(grid, B)] + _pad_to([grid.stride(1 + d) for d in range(dim)], 3) + [grid.stride(-1)]


class _PackedLattice:
    """A SeparableGrid prepared for the C-ABI: the packed coordinate vectors + the shape the
    host code sees, (1, *out, D).  Adds INTERPOL_FLAG_SEPARABLE_GRID to the call."""

    def __init__(self, sep, gdt):
        self.buf = sep.packed(gdt)
        self.shape = sep.shape

    def data_ptr(self):
        return self.buf.data_ptr()

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _prep_grid(grid, gdt):
    """-> (grid object to pass on, extra flags)."""
    if isinstance(grid, LazyGrid):
        return _PackedLattice(grid, gdt), (FLAG_AFFINE_GRID if isinstance(grid, AffineGrid) else FLAG_SEPARABLE_GRID)
    grid = grid.to(gdt)
    if not _spatially_contiguous(grid, 1):
        grid = grid.contiguous()
    return grid, 0


def gather(op, vol, grid, bound, order, extrapolate, flags=0, out=None):
    """pull / grad / hess: vol (B,C,*in), grid (B,*out,D) -> val (B,C,*out[,D[,D]]).
    `out`: a dense tensor of that shape and dtype to write into instead of allocating one."""
    dev = _require_gpu(vol, grid)
    dim = grid.shape[-1]
    if dim not in (1, 2, 3):
        raise NotImplementedError("interpol (MI355X build): only 1-D, 2-D and 3-D grids are supported")
    dt, gdt = common_dtypes(vol, grid)
    out_dt = dt
    if op in ("hess",) and dt in (torch.bfloat16, torch.float16):
        dt = torch.float32                                 # second-order operator: f32 / f64 kernels only
    vol = vol.to(dt)
    grid, gflag = _prep_grid(grid, gdt)
    flags |= gflag
    routed = False
    lin3 = op == "pull" and dim == 3 and dt in (torch.bfloat16, torch.float16) and gdt == torch.float32 and all(int(o) == 1 for o in order[:3])   # trilinear: routed in 16 bits too
    if (((op in ("pull", "grad") and dim == 3 and dt == torch.float32) or lin3 or (op == "pull" and dim == 2 and dt in _ROUTED_2D and gdt == torch.float32))
            and not (flags & (FLAG_NO_FASTPATH | FLAG_FORCE_TILED | FLAG_BINNED_SCATTER)) and (flags >> 8) == 0):
        # the router of the pull (csrc/push_owner.hip: own_gather; 2-D: csrc/scatter2d.hip: gather2d): like the push's, see interpol/backend.py
        from . import backend
        if backend.rough_deformations is None:
            flags |= FLAG_AUTO_SCATTER
        elif backend.rough_deformations:
            flags |= FLAG_BINNED_SCATTER
    if op in ("pull", "grad") and (flags & (FLAG_AUTO_SCATTER | FLAG_BINNED_SCATTER)):
        routed = True
    B = max(vol.shape[0], grid.shape[0])
    C = vol.shape[1]
    oshape = list(grid.shape[1:-1])
    trailing = {"pull": [], "grad": [dim], "hess": [dim, dim]}[op]
    if out is None:
        val = torch.empty([B, C] + oshape + trailing, dtype=dt, device=dev)
    else:
        val = out
        if not (val.is_contiguous() and val.dtype == dt and list(val.shape) == [B, C] + oshape + trailing):
            raise ValueError("gather output: expected a contiguous %s tensor of shape %s, got %s %s"
                             % (dt, [B, C] + oshape + trailing, val.dtype, list(val.shape)))
    if val.numel() == 0:
        return val.to(out_dt)
    vstr = [_bstride(vol, B), vol.stride(1)] + _pad_to([vol.stride(2 + d) for d in range(dim)], 3)
    valstr = [val.stride(0), val.stride(1)] + _pad_to([val.stride(2 + d) for d in range(dim)], 3) + [0, 0]
    p = make_problem(dim, dt, gdt, bound, order, extrapolate, B, C, vol.shape[2:], oshape,
                     vstr, _grid_strides(grid, B, dim), valstr, flags)
    L = lib()
    if routed:
        # workspace of the routed pull (18 B per sample + 2 KiB per brick of the image; 0: the organisation does not apply).  When
        # it cannot be allocated the call is the plain interpol_pull: the sample tiles need none.
        wbytes = int(L.interpol_pull_workspace(ctypes.byref(p)))
        ws = _optional_workspace(wbytes, dev)
        if ws is not None:
            if _POISON_SCRATCH:
                ws.fill_(0xff)
            with torch.cuda.device(dev):
                rc = (L.interpol_pull_ws if op == "pull" else L.interpol_grad_ws)(ctypes.byref(p), _ptr(vol), _ptr(grid), _ptr(val), _ptr(ws), wbytes, _stream(dev))
            _check(rc, "interpol_%s_ws" % op)
            return val.to(out_dt)
        p.flags &= ~(FLAG_AUTO_SCATTER | FLAG_BINNED_SCATTER)
    fn = getattr(L, "interpol_" + op)
    with torch.cuda.device(dev):
        rc = fn(ctypes.byref(p), _ptr(vol), _ptr(grid), _ptr(val), _stream(dev))
    _check(rc, "interpol_" + op)
    return val.to(out_dt)


def scatter(op, val, grid, shape, bound, order, extrapolate, flags=0, out=None, shared=False, with_count=False, need_workspace=False):
    """push / count / pushgrad: val (B,C,*in[,D]) , grid (B,*in,D) -> vol (B,C,*shape).
    `out` (dense, same dtype) + FLAG_ACCUMULATE adds into an existing target.
    `shared=True`: ONE target (1,C,*shape) that all batch items accumulate into
    (the reference's grid_push(...).sum(0), without the B per-item volumes).
    `with_count=True` (push only): the target has C + 1 channels and channel C receives the count
    image of the same grid, splatted in the same pass (INTERPOL_FLAG_WITH_COUNT).
    `need_workspace=True`: return None (nothing launched) when the probe-routed organisation's workspace is denied, so that the caller
    can pick another organisation instead of the tiles."""
    dev = _require_gpu(val, grid)
    dim = grid.shape[-1]
    if dim not in (1, 2, 3):
        raise NotImplementedError("interpol (MI355X build): only 1-D, 2-D and 3-D grids are supported")
    from . import backend
    if backend.want_exact_scatter():
        flags |= FLAG_NO_FASTPATH                       # float atomics, like the reference's scatter_add_
    elif op in ("push", "count") and not (flags & (FLAG_NO_FASTPATH | FLAG_FORCE_TILED | FLAG_BINNED_SCATTER)) and (flags >> 8) == 0:
        if backend.rough_deformations is None:
            flags |= FLAG_AUTO_SCATTER                  # a probe of this call picks tiles or owner-computes (csrc/push_owner.hip)
        elif backend.rough_deformations:
            flags |= FLAG_BINNED_SCATTER                # owner-computes organisation, always
    dt, gdt = common_dtypes(val, grid)
    out_dt = dt
    if op == "pushgrad" and dt in (torch.bfloat16, torch.float16):
        dt = torch.float32
    grid, gflag = _prep_grid(grid, gdt)
    flags |= gflag
    gshape = list(grid.shape[1:-1])
    if shape is None:
        shape = gshape
    shape = [int(s) for s in shape]
    if op == "count":
        B, C = grid.shape[0], 1
        valstr = [0] * 7
    else:
        val = val.to(dt)
        if not _spatially_contiguous(val, 2):
            val = val.contiguous()
        B = max(val.shape[0], grid.shape[0])
        C = val.shape[1]
        valstr = [_bstride(val, B), val.stride(1)] + _pad_to([val.stride(2 + d) for d in range(dim)], 3) + [0, 0]
    Bv = 1 if shared else B
    if with_count:
        if op != "push":
            raise ValueError("with_count applies to push only")
        flags |= FLAG_WITH_COUNT
    Cv = C + (1 if with_count else 0)
    if out is None:
        vol = torch.empty([Bv, Cv] + shape, dtype=dt, device=dev)
    else:
        vol = out
        if not (vol.is_contiguous() and vol.dtype == dt and list(vol.shape) == [Bv, Cv] + shape):
            raise ValueError("scatter target: expected a contiguous %s tensor of shape %s, got %s %s"
                             % (dt, [Bv, Cv] + shape, vol.dtype, list(vol.shape)))
        if (flags & FLAG_ACCUMULATE) and dt in (torch.bfloat16, torch.float16):
            raise ValueError("FLAG_ACCUMULATE needs a float32 / float64 target (a low-precision target is narrowed once, not accumulated)")
    if vol.numel() == 0:
        return vol.to(out_dt)
    if grid.numel() == 0:
        return vol.zero_().to(out_dt) if out is None else vol
    vstr = [0 if shared else vol.stride(0), vol.stride(1)] + _pad_to([vol.stride(2 + d) for d in range(dim)], 3)
    p = make_problem(dim, dt, gdt, bound, order, extrapolate, B, C, shape, gshape,
                     vstr, _grid_strides(grid, B, dim), valstr, flags)
    L = lib()
    # scratch: the fp32 accumulator of a 16-bit target, followed by the workspace of the binned
    # organisation when the library wants one for this problem (interpol_hip.h)
    scratch, sbytes = None, 0
    if op in ("push", "count"):
        sbytes = int(L.interpol_scatter_workspace(ctypes.byref(p), 1 if op == "count" else 0))
    if dt in (torch.bfloat16, torch.float16):
        sbytes = max(sbytes, vol.numel() * 4)
    if sbytes > 0:
        if flags & FLAG_AUTO_SCATTER:
            # The probe-routed default asks for the owner-computes workspace (~22 B per sample + 1 KiB per brick, api.py)
            # whether or not the probe will pick that organisation; when it does not fit comfortably (_optional_workspace), the
            # call falls back to the tiles, which need none -- a push that fitted without the router still fits.
            scratch = _optional_workspace(sbytes, dev)
            if scratch is None and need_workspace:
                return None
            if scratch is None:
                flags &= ~FLAG_AUTO_SCATTER
                p = make_problem(dim, dt, gdt, bound, order, extrapolate, B, C, shape, gshape,
                                 vstr, _grid_strides(grid, B, dim), valstr, flags)
                sbytes = int(L.interpol_scatter_workspace(ctypes.byref(p), 1 if op == "count" else 0))
                if dt in (torch.bfloat16, torch.float16):
                    sbytes = max(sbytes, vol.numel() * 4)
                scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev) if sbytes > 0 else None
        else:
            scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        if scratch is not None and _POISON_SCRATCH:
            scratch.fill_(0xff)                          # (debugging aid: INTERPOL_POISON_SCRATCH=1 -- the kernels must not depend on stale workspace contents)
    with torch.cuda.device(dev):
        if op == "count":
            rc = L.interpol_count(ctypes.byref(p), _ptr(grid), _ptr(vol), _ptr(scratch), sbytes, _stream(dev))
        else:
            rc = getattr(L, "interpol_" + op)(ctypes.byref(p), _ptr(val), _ptr(grid), _ptr(vol),
                                              _ptr(scratch), sbytes, _stream(dev))
    _check(rc, "interpol_" + op)
    return vol.to(out_dt)


def pull_backward(gout, vol, grid, bound, order, extrapolate, need_vol, need_grid, flags=0):
    """Fused backward of pull: (grad_vol (B,C,*in) | None, grad_grid (B,*out,D) | None)."""
    dev = _require_gpu(gout, vol, grid)
    dim = grid.shape[-1]
    from . import backend
    if (need_vol and flags == 0 and dim == 3 and torch.is_tensor(grid) and len(set(order[:3])) == 1
            and order[0] in (0, 1, 2, 3)
            and vol.dtype == torch.float32 and gout.dtype == torch.float32 and grid.dtype == torch.float32
            and vol.shape[0] == grid.shape[0] == gout.shape[0] and not backend.want_exact_scatter()
            and backend.rough_deformations is not False):
        # 3-D quadratic / cubic: the image gradient IS grid_push of grad_out (pushpull.py:252-255) and the library computes it
        # with the push kernels anyway; through `scatter` it also gets the push's routing (rough / folding fields take the
        # owner-computes organisation: 31 -> 10 ms for the backward of a field of 8 voxels amplitude)
        # (trilinear and nearest neighbour as well -- trilinear, sigma = 6: 15.9 -> 3.1 ms; nearest, sigma = 2: 5.7 -> 2.5 ms.  The library
        # splits every backward since round 5: its fused tile kernel for both gradients was wrong under rough fields)
        gvol = scatter("push", gout, grid, list(vol.shape[2:]), bound, order, extrapolate)
        ggrid = pull_backward(gout, vol, grid, bound, order, extrapolate, False, True, flags)[1] if need_grid else None
        return gvol, ggrid
    if (need_vol and flags == 0 and dim == 2 and torch.is_tensor(grid) and 1 <= min(order[:2]) and max(order[:2]) <= 3
            and vol.dtype == gout.dtype and vol.dtype in _ROUTED_2D and grid.dtype == torch.float32
            and vol.shape[0] == grid.shape[0] == gout.shape[0]
            and not backend.want_exact_scatter() and backend.rough_deformations is not False):
        # 2-D, orders 1..3: the library splits this backward into a push of grad_out and the contracted grid gradient anyway
        # (csrc/abi.hip: interpol_pull_backward); through `scatter` the push gets its router (csrc/scatter2d.hip), and so does
        # the grid gradient below
        gvol = scatter("push", gout, grid, list(vol.shape[2:]), bound, order, extrapolate)
        ggrid = pull_backward(gout, vol, grid, bound, order, extrapolate, False, True, flags)[1] if need_grid else None
        return gvol, ggrid
    if (need_vol and flags == 0 and dim == 3 and torch.is_tensor(grid) and max(order[:3]) <= 3 and max(order[:3]) > 0
            and vol.dtype == gout.dtype == grid.dtype == torch.float64 and vol.shape[0] == grid.shape[0] == gout.shape[0]
            and grid[0, ..., 0].numel() >= 4096 and not backend.want_exact_scatter()):
        # float64, orders <= 3: the same split -- the image gradient through the float64 LDS tiles of grid_push (csrc/push_f64.hip)
        # instead of one scattered global atomic per tap inside the fused kernel (1 x 2 x 128^3, sigma = 2: 12.7 -> 2 ms)
        gvol = scatter("push", gout, grid, list(vol.shape[2:]), bound, order, extrapolate)
        ggrid = pull_backward(gout, vol, grid, bound, order, extrapolate, False, True, flags)[1] if need_grid else None
        return gvol, ggrid
    if backend.want_exact_scatter() and need_vol:
        flags |= FLAG_NO_FASTPATH                       # grad_vol is a scatter
    dt, gdt = common_dtypes(vol, grid)
    vol = vol.to(dt)
    gout = gout.to(dt)
    grid_c, gflag = _prep_grid(grid, gdt)
    flags |= gflag
    if gflag and need_grid:
        raise RuntimeError("interpol: a SeparableGrid / AffineGrid is a constant lattice, it has no gradient")
    if not _spatially_contiguous(vol, 2):
        vol = vol.contiguous()
    if not _spatially_contiguous(gout, 2):
        gout = gout.contiguous()
    B = max(vol.shape[0], grid_c.shape[0])
    C = vol.shape[1]
    ishape = list(vol.shape[2:])
    oshape = list(grid_c.shape[1:-1])
    gvol = torch.empty([B, C] + ishape, dtype=dt, device=dev) if need_vol else None
    ggrid = torch.empty([B] + oshape + [dim], dtype=gdt, device=dev) if need_grid else None
    if gout.numel() == 0:
        return (gvol.zero_() if need_vol else None), ggrid
    scratch, sbytes = None, 0
    if need_vol and dt in (torch.bfloat16, torch.float16):
        scratch = torch.empty(gvol.numel(), dtype=torch.float32, device=dev)
        sbytes = scratch.numel() * 4
    vstr = [_bstride(vol, B), vol.stride(1)] + _pad_to([vol.stride(2 + d) for d in range(dim)], 3)
    valstr = [_bstride(gout, B), gout.stride(1)] + _pad_to([gout.stride(2 + d) for d in range(dim)], 3) + [0, 0]
    routed = 0
    high = dim == 3 and all(int(o) == int(order[0]) for o in order[:3]) and int(order[0]) in (4, 5)   # image gradient through scatter5
    two_d = dim == 2 and need_grid and dt in _ROUTED_2D and (not need_vol or dt == torch.float32)   # (a 16-bit image gradient keeps `scratch` for its accumulator)
    if ((((need_grid or (need_vol and high)) and dim == 3 and dt == torch.float32) or two_d) and gdt == torch.float32 and (flags >> 8) == 0
            and not (flags & (FLAG_NO_FASTPATH | FLAG_FORCE_TILED | FLAG_BINNED_SCATTER))):
        # the grid gradient takes the router of the pull (csrc/push_owner.hip: own_gather<K, true>): tiles whose samples leave the
        # LDS box go to the bricks of the image; the workspace rides in the `scratch` argument (interpol_hip.h)
        routed = FLAG_AUTO_SCATTER if backend.rough_deformations is None else (FLAG_BINNED_SCATTER if backend.rough_deformations else 0)
    elif (need_grid or (need_vol and high)) and (flags & FLAG_BINNED_SCATTER) and (dim == 3 or two_d):
        routed = FLAG_BINNED_SCATTER
    p = make_problem(dim, dt, gdt, bound, order, extrapolate, B, C, ishape, oshape,
                     vstr, _grid_strides(grid_c, B, dim), valstr, flags | routed)
    if routed:
        wbytes = int(lib().interpol_pull_workspace(ctypes.byref(p)))
        scratch = _optional_workspace(wbytes, dev)
        sbytes = wbytes if scratch is not None else 0
        if scratch is None:
            p.flags &= ~(FLAG_AUTO_SCATTER | FLAG_BINNED_SCATTER)
    with torch.cuda.device(dev):
        rc = lib().interpol_pull_backward(ctypes.byref(p), _ptr(gout), _ptr(vol), _ptr(grid_c), _ptr(gvol),
                                          _ptr(ggrid), _ptr(scratch), sbytes, _stream(dev))
    _check(rc, "interpol_pull_backward")
    return gvol, ggrid


def push_backward(gvol_out, val, grid, bound, order, extrapolate, need_val, need_grid, flags=0):
    """Fused backward of push (val given) or count (val None):
    (grad_val (B,C,*in) | None, grad_grid (B,*in,D) | None)."""
    dev = _require_gpu(gvol_out, val, grid)
    dim = grid.shape[-1]
    dt, gdt = common_dtypes(gvol_out, grid)
    gvol_out = gvol_out.to(dt)
    grid_c, gflag = _prep_grid(grid, gdt)
    flags |= gflag
    if gflag and need_grid:
        raise RuntimeError("interpol: a SeparableGrid / AffineGrid is a constant lattice, it has no gradient")
    gshape = list(grid_c.shape[1:-1])
    B = max(gvol_out.shape[0], grid_c.shape[0])
    C = gvol_out.shape[1]
    count = val is None
    if not count:
        val = val.to(dt)
        if not val.is_contiguous():
            val = val.contiguous()
        B = max(B, val.shape[0])
        if val.shape[0] != B:
            val = val.expand([B] + list(val.shape[1:])).contiguous()
    gval = torch.empty([B, C] + gshape, dtype=dt, device=dev) if (need_val and not count) else None
    ggrid = torch.empty([B] + gshape + [dim], dtype=gdt, device=dev) if need_grid else None
    if grid_c.numel() == 0 or (gval is None and ggrid is None):
        return gval, ggrid
    vstr = [_bstride(gvol_out, B), gvol_out.stride(1)] + _pad_to([gvol_out.stride(2 + d) for d in range(dim)], 3)
    dense = _dense_strides([B, C] + gshape)
    valstr = dense[:2] + _pad_to(dense[2:], 3) + [0, 0]
    p = make_problem(dim, dt, gdt, bound, order, extrapolate, B, C, gvol_out.shape[2:], gshape,
                     vstr, _grid_strides(grid_c, B, dim), valstr, flags)
    L = lib()
    from . import backend
    routed = 0
    if (((dim == 3 and dt == torch.float32) or (dim == 2 and dt in _ROUTED_2D)) and gdt == torch.float32 and (flags >> 8) == 0
            and not (flags & (FLAG_NO_FASTPATH | FLAG_FORCE_TILED | FLAG_BINNED_SCATTER))):
        # both gradients are gathers: they take the router of the pull (bricks of the image, csrc/push_owner.hip: own_gather)
        routed = FLAG_AUTO_SCATTER if backend.rough_deformations is None else (FLAG_BINNED_SCATTER if backend.rough_deformations else 0)
    elif flags & FLAG_BINNED_SCATTER:
        routed = FLAG_BINNED_SCATTER
    if routed:
        p.flags |= routed
        wbytes = int(L.interpol_pull_workspace(ctypes.byref(p)))
        ws = _optional_workspace(wbytes, dev)
        if ws is not None:
            with torch.cuda.device(dev):
                rc = L.interpol_push_backward_ws(ctypes.byref(p), _ptr(gvol_out), _ptr(None if count else val), _ptr(grid_c),
                                                 _ptr(gval), _ptr(ggrid), _ptr(ws), wbytes, _stream(dev))
            _check(rc, "interpol_push_backward_ws")
            return gval, ggrid
        p.flags &= ~(FLAG_AUTO_SCATTER | FLAG_BINNED_SCATTER)
    with torch.cuda.device(dev):
        if count:
            rc = L.interpol_count_backward(ctypes.byref(p), _ptr(gvol_out), _ptr(grid_c), _ptr(ggrid), _stream(dev))
        else:
            rc = L.interpol_push_backward(ctypes.byref(p), _ptr(gvol_out), _ptr(val), _ptr(grid_c),
                                          _ptr(gval), _ptr(ggrid), _stream(dev))
    _check(rc, "interpol_push_backward")
    return gval, ggrid


def spline_filter_(data, bound, order, dim, src=None):
    """Prefilter of `data` (contiguous) along dimension `dim`: in place, or -- `src` given, same shape
    and dtype, contiguous, not overlapping -- reading `src` and writing `data`."""
    dev = _require_gpu(data) if src is None else _require_gpu(data, src)
    if data.dtype not in _DTYPE_CODE:
        raise TypeError("spline_coeff: unsupported dtype %s" % data.dtype)
    if not data.is_contiguous():
        raise ValueError("spline_filter_: `data` must be contiguous")
    if src is not None and (src.shape != data.shape or src.dtype != data.dtype or not src.is_contiguous()):
        raise ValueError("spline_filter_: `src` must match `data` (shape, dtype, contiguous)")
    dim = dim % data.dim()
    n = data.shape[dim]
    outer = 1
    for s in data.shape[:dim]:
        outer *= s
    inner = 1
    for s in data.shape[dim + 1:]:
        inner *= s
    with torch.cuda.device(dev):
        if src is None:
            rc = lib().interpol_spline_filter(_ptr(data), _DTYPE_CODE[data.dtype], outer, n, inner,
                                              int(bound), int(order), _stream(dev))
        else:
            rc = lib().interpol_spline_filter_to(_ptr(src), _ptr(data), _DTYPE_CODE[data.dtype], outer, n, inner,
                                                 int(bound), int(order), _stream(dev))
    _check(rc, "interpol_spline_filter")
    return data


def resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint=False, n_lattice=None):
    """One pass of a tensor-product resampling along `dim` (interpol_resample_1d).
    forward: src (..., n_lattice, ...), lin (n_samples,) -> (..., n_samples, ...)
    adjoint: src (..., n_samples, ...) -> (..., n_lattice, ...)   (f32 / f64)."""
    dev = _require_gpu(src, lin)
    if src.dtype not in _DTYPE_CODE:
        raise TypeError("resample1d: unsupported dtype %s" % src.dtype)
    src = src.contiguous()
    ldt = torch.float64 if src.dtype == torch.float64 else torch.float32
    lin = lin.detach().to(ldt).contiguous()
    dim = dim % src.dim()
    n = src.shape[dim]
    outer = 1
    for s_ in src.shape[:dim]:
        outer *= s_
    inner = 1
    for s_ in src.shape[dim + 1:]:
        inner *= s_
    if adjoint:
        if n != lin.numel():
            raise ValueError("resample1d: the resampled dimension must match the lattice vector")
        ns, nl = n, int(n_lattice)
        oshape = list(src.shape[:dim]) + [nl] + list(src.shape[dim + 1:])
    else:
        ns, nl = lin.numel(), n
        oshape = list(src.shape[:dim]) + [ns] + list(src.shape[dim + 1:])
    dst = torch.empty(oshape, dtype=src.dtype, device=dev)
    if dst.numel() == 0:
        return dst
    if src.numel() == 0:
        return dst.zero_()
    with torch.cuda.device(dev):
        rc = lib().interpol_resample_1d(_DTYPE_CODE[src.dtype], _DTYPE_CODE[ldt], int(order), int(bound), int(extrapolate),
                                        int(mode), int(bool(adjoint)), outer, ns, nl, inner,
                                        _ptr(src), _ptr(lin), _ptr(dst), _stream(dev))
    _check(rc, "interpol_resample_1d")
    return dst


def labels_covered(dim, order):
    """Does interpol_pull_labels take this stencil?  (all dims one order <= 3: up to the 64 taps of the 3-D cubic)"""
    order = list(order)[:dim]
    return len(set(order)) == 1 and order[0] <= 3


def pull_labels(vol, grid, bound, order, extrapolate, flags=0):
    """Label-map pull: vol (B,C,*in) integer labels, grid (B,*out,D) float32 -> (B,C,*out) int32.
    Arg-max over the labels of the interpolated indicator images (reference api.py:194-205)."""
    dev = _require_gpu(vol, grid)
    dim = grid.shape[-1]
    if vol.dtype.is_floating_point:
        raise TypeError("pull_labels: integer label map expected")
    if grid.dtype != torch.float32:
        raise TypeError("pull_labels: float32 coordinates expected")
    vol = vol.to(torch.int32)
    grid, gflag = _prep_grid(grid, torch.float32)
    flags |= gflag
    B = max(vol.shape[0], grid.shape[0])
    C = vol.shape[1]
    oshape = list(grid.shape[1:-1])
    val = torch.empty([B, C] + oshape, dtype=torch.int32, device=dev)
    if val.numel() == 0:
        return val
    vstr = [_bstride(vol, B), vol.stride(1)] + _pad_to([vol.stride(2 + d) for d in range(dim)], 3)
    valstr = [val.stride(0), val.stride(1)] + _pad_to([val.stride(2 + d) for d in range(dim)], 3) + [0, 0]
    p = make_problem(dim, torch.float32, torch.float32, bound, order, extrapolate, B, C, vol.shape[2:], oshape,
                     vstr, _grid_strides(grid, B, dim), valstr, flags)
    with torch.cuda.device(dev):
        rc = lib().interpol_pull_labels(ctypes.byref(p), _ptr(vol), _ptr(grid), _ptr(val), _stream(dev))
    _check(rc, "interpol_pull_labels")
    return val


def bricks_applicable(val, grid, with_count, order=None):
    """Can interpol_push_bricks take this problem? (3-D, fp32, one order <= 3 for all dims,
    <= 4 target channels, < 2^32 samples)"""
    dim = grid.shape[-1]
    from . import backend
    if backend.want_exact_scatter():                     # the brick kernels accumulate in fixed point too
        return False
    if order is not None:
        o = list(order)[:3]
        if len(set(o)) != 1 or o[0] > 3:
            return False
    return (dim == 3 and val is not None and val.dtype == torch.float32 and grid.dtype == torch.float32
            and val.shape[1] + (1 if with_count else 0) <= 4
            and max(val.shape[0], grid.shape[0]) * int(torch.Size(grid.shape[1:-1]).numel()) < 2 ** 32)


def push_bricks(val, grid, shape, bound, order, extrapolate, flags=0, out=None, shared=False, with_count=False):
    """interpol_push_bricks: push (and count) organised by target brick -- for expanding deformations.
    Same arguments and result as `scatter("push", ...)`."""
    dev = _require_gpu(val, grid)
    dim = grid.shape[-1]
    if not bricks_applicable(val, grid, with_count, order):
        raise ValueError("push_bricks: 3-D float32 problems, one order <= 3, at most 4 target channels only")
    grid, gflag = _prep_grid(grid, torch.float32)
    flags |= gflag | (FLAG_WITH_COUNT if with_count else 0)
    val = val.contiguous()
    gshape = list(grid.shape[1:-1])
    shape = [int(s) for s in (gshape if shape is None else shape)]
    B = max(val.shape[0], grid.shape[0])
    C = val.shape[1]
    Cv = C + (1 if with_count else 0)
    Bv = 1 if shared else B
    if out is None:
        vol = torch.empty([Bv, Cv] + shape, dtype=torch.float32, device=dev)
    else:
        vol = out
        assert vol.is_contiguous() and vol.dtype == torch.float32 and list(vol.shape) == [Bv, Cv] + shape
    if vol.numel() == 0:
        return vol
    if grid.numel() == 0:
        return vol.zero_() if out is None else vol
    valstr = [_bstride(val, B), val.stride(1)] + _pad_to([val.stride(2 + d) for d in range(dim)], 3) + [0, 0]
    vstr = [0 if shared else vol.stride(0), vol.stride(1)] + _pad_to([vol.stride(2 + d) for d in range(dim)], 3)
    p = make_problem(dim, torch.float32, torch.float32, bound, order, extrapolate, B, C, shape, gshape,
                     vstr, _grid_strides(grid, B, dim), valstr, flags)
    L = lib()
    nbytes = L.interpol_push_bricks_workspace(ctypes.byref(p))
    if nbytes < 0:
        _check(int(nbytes), "interpol_push_bricks_workspace")
    work = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.interpol_push_bricks(ctypes.byref(p), _ptr(val), _ptr(grid), _ptr(vol), _ptr(work), int(nbytes), _stream(dev))
    _check(rc, "interpol_push_bricks")
    return vol


HANDBACK_MODES = {"adaptive": 0, "always": 1, "never": 2}


def set_handback(mode):
    """Tile hand-back policy of the library (include/interpol_hip.h, interpol_set_handback): 'adaptive' (default: per
    stream, from the flags of its recent launches -- results may differ in the last bits with the stream's history),
    'always' or 'never' (every operator is then a deterministic function of its inputs).  Returns the previous mode."""
    code = HANDBACK_MODES[mode] if isinstance(mode, str) else int(mode)
    prev = int(lib().interpol_set_handback(code))
    return {v: k for k, v in HANDBACK_MODES.items()}.get(prev, prev)


def release_stream(stream=None):
    """Give the hand-back slot of `stream` (a torch.cuda.Stream; default: the current one) back to the library."""
    st = torch.cuda.current_stream() if stream is None else stream
    with torch.cuda.device(st.device):
        return bool(lib().interpol_release_stream(ctypes.c_void_p(st.cuda_stream)))
