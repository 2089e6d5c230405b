"""Device-generic kernel table: the operators of `interpol/ops.py` in plain PyTorch.

The HIP library (`_hip.py`) covers D in {1, 2, 3} on the GPU.  Everything else the reference
accepts -- any number of spatial dims through its `nd` path (`pushpull.py:49-66`, `nd.py:80-464`)
and tensors that live on the CPU -- is served here, so that the package stays a drop-in.  This is
product code (the oracle under `oracle/` is test infrastructure and is never imported here); it is
NOT the fast path and is never taken for CUDA tensors of 1-3 spatial dims.

Formulation (not the reference's (K+1)^D sequential passes): per chunk of sample points the whole
stencil is materialised at once -- linear tap indices and tap weights of shape (B, n, T),
T = prod_d (K_d + 1) -- then ONE gather + weighted sum (pull / grad / hess) or ONE `scatter_add_`
(push / count / pushgrad).  Chunks bound the temporary to ~2^24 elements.

Numerical definition: stencil `nd.py:44-61`; weights = centred cardinal B-splines (`splines.py:30-195`),
evaluated with the Cox - de Boor recursion (equal to the reference's Horner forms up to rounding);
boundary conditions `bounds.py:30-89`; extrapolation mask `nd.py:10-27`; `iso0` / `iso1` semantics
for all-nearest / all-linear problems of D <= 3 (`pushpull.py:50-64`); backward compositions
`pushpull.py:237-325`.  Reference quirks kept: B-3 (dst1 sign 0 at index 0) and B-4 (order-1
gradient sign in the nd path); B-1 / B-2 follow the intended semantics, as the HIP kernels do.
"""
import torch

_CHUNK_ELEMS = 1 << 24


# ---------------------------------------------------------------------------
# boundary conditions (bounds.py:30-89): codes 0 zero, 1 replicate, 2 dct1, 3 dct2, 4 dst1, 5 dst2, 6 dft
# ---------------------------------------------------------------------------
def _pymod(a, m):
    return torch.remainder(a, m)


def bound_index(bound, i, n):
    """Bound.index: lattice index of tap position i (int64 tensor) on a lattice of n points."""
    if bound in (0, 1):
        return i.clamp(0, n - 1)
    if bound == 2:
        if n == 1:
            return torch.zeros_like(i)
        m = 2 * (n - 1)
        j = _pymod(i.abs(), m)
        return torch.where(j >= n, m - j, j)
    if bound in (3, 5):
        m = 2 * n
        j = torch.where(i < 0, m - 1 - _pymod(-i - 1, m), _pymod(i, m))
        return torch.where(j >= n, m - 1 - j, j)
    if bound == 4:
        m = 2 * (n + 1)
        j = _pymod(torch.where(i < 0, -i - 2, i), m)
        j = torch.where(j > n, m - 2 - j, j)
        j = torch.where(j == -1, torch.zeros_like(j), j)
        return torch.where(j == n, torch.full_like(j, n - 1), j)
    if bound == 6:
        return _pymod(i, n)
    raise ValueError('Unknown boundary condition code {}'.format(bound))


def bound_sign(bound, i, n):
    """Bound.transform as a multiplicative factor (None -> no tensor: returns None)."""
    if bound == 0:
        return ((i >= 0) & (i < n)).to(torch.int8)
    if bound == 5:
        j = torch.where(i < 0, n - 1 - i, i)
        odd = (torch.div(j, n, rounding_mode='floor') % 2) == 1
        return torch.where(odd, -torch.ones_like(i), torch.ones_like(i)).to(torch.int8)
    if bound == 4:
        if n == 1:
            return None
        m = 2 * (n + 1)
        j = _pymod(torch.where(i < 0, -i + (n - 1), i), m)
        s = (j != 0) & (_pymod(j, n + 1) != n)                       # quirk B-3: 0 at j == 0
        odd = (torch.div(j, n + 1, rounding_mode='floor') % 2) == 1
        return torch.where(odd, -s.to(torch.int8), s.to(torch.int8))
    return None


# ---------------------------------------------------------------------------
# B-spline weights (splines.py:30-195)
# ---------------------------------------------------------------------------
def _bspline(order, x):
    """beta^order(x), Cox - de Boor; exact on the support, 0 outside."""
    if order == 0:
        return torch.ones_like(x)
    if order == 1:
        return (1 - x.abs()).clamp_min(0)
    h = 0.5 * (order + 1)
    return ((x + h) * _bspline(order - 1, x + 0.5) + (h - x) * _bspline(order - 1, x - 0.5)) / order


def _bspline0(x):
    # the order-0 factor of the recursion for derivatives: indicator of [-1/2, 1/2)
    return ((x >= -0.5) & (x < 0.5)).to(x.dtype)


def _bspline_rec(order, x):
    """beta^order with the TRUE order-0 base case (needed below order 1 by the derivative recursions)."""
    if order == 0:
        return _bspline0(x)
    return _bspline(order, x)


def _dbspline(order, x, nd_quirk):
    """d/dx beta^order(x)  (fastgrad).  Order 1 in the reference's nd path returns +sign(x) (quirk B-4)."""
    if order == 0:
        return torch.zeros_like(x)
    if order == 1:
        return torch.sign(x) if nd_quirk else -torch.sign(x)
    return _bspline_rec(order - 1, x + 0.5) - _bspline_rec(order - 1, x - 0.5)


def _d2bspline(order, x):
    """d2/dx2 beta^order(x)  (fasthess; zero for orders 0 and 1)."""
    if order < 2:
        return torch.zeros_like(x)
    if order == 2:
        # piecewise constant: the reference's piece choice at the breakpoints (splines.py:157-158)
        return torch.where(x.abs() < 0.5, torch.full_like(x, -2.0), torch.ones_like(x))
    return _dbspline(order - 1, x + 0.5, False) - _dbspline(order - 1, x - 0.5, False)


# ---------------------------------------------------------------------------
# the stencil of a chunk of samples
# ---------------------------------------------------------------------------
def _mode(order, dim):
    if dim <= 3 and all(o == 1 for o in order):
        return 'iso1'
    if dim <= 3 and all(o == 0 for o in order):
        return 'iso0'
    return 'nd'


class _Stencil:
    """Per dim d: idx[d] (B, n, K_d + 1) int64, w / g / h[d] same shape (sign folded in)."""

    def __init__(self, coords, shape, bound, order, need):
        dim = coords.shape[-1]
        mode = _mode(order, dim)
        self.idx, self.w, self.g, self.h = [], [], [], []
        for d in range(dim):
            x = coords[..., d]
            k = order[d]
            if mode == 'iso0':
                i0 = torch.round(x)                                  # iso0.py:12: half to even
            else:
                i0 = torch.floor(x - 0.5 * (k - 1))                  # nd.py:45 (iso1.py:13 for k = 1)
            t = x - i0                                               # nd.py:46
            i0 = i0.clamp(-2.0 ** 62, 2.0 ** 62).long()
            taps = torch.arange(k + 1, device=x.device)
            pos = i0[..., None] + taps
            xj = t[..., None] - taps.to(t.dtype)
            ii = bound_index(bound[d], pos, shape[d])
            sg = bound_sign(bound[d], pos, shape[d])
            if mode == 'iso1':
                w = torch.stack([1 - t, t], -1)                      # iso1.py:19-20
                g = torch.stack([-torch.ones_like(t), torch.ones_like(t)], -1) if need >= 1 else None
                h = torch.zeros_like(w) if need >= 2 else None
            else:
                w = _bspline(k, xj)
                g = _dbspline(k, xj, mode == 'nd') if need >= 1 else None
                h = _d2bspline(k, xj) if need >= 2 else None
            if sg is not None:
                sf = sg.to(w.dtype)
                w = w * sf
                g = g * sf if g is not None else None
                h = h * sf if h is not None else None
            self.idx.append(ii); self.w.append(w); self.g.append(g); self.h.append(h)
        self.dim = dim

    def _outer(self, factors):
        out = None
        for f in factors:
            out = f if out is None else (out[..., :, None] * f[..., None, :]).flatten(-2)
        return out

    def linear_index(self, shape):
        out, stride = None, 1
        strides = []
        for n in reversed(shape):
            strides.append(stride); stride *= n
        strides = strides[::-1]
        for d in range(self.dim):
            term = self.idx[d] * strides[d]
            out = term if out is None else (out[..., :, None] + term[..., None, :]).flatten(-2)
        return out

    def weights(self, deriv=()):
        """product of per-dim factors; dims listed in `deriv` use the gradient (once) or the hessian (twice)."""
        fs = []
        for d in range(self.dim):
            c = deriv.count(d)
            fs.append(self.w[d] if c == 0 else (self.g[d] if c == 1 else self.h[d]))
        return self._outer(fs)


def _mask(coords, shape, extrapolate):
    """nd.py:10-27: 1 inside the field of view (with tolerance), None when everything is kept."""
    if extrapolate == 1:
        return None
    thr = 0.05 if extrapolate == 0 else 0.55
    m = None
    for d, n in enumerate(shape):
        x = coords[..., d]
        md = (x > -thr) & (x < n - 1 + thr)
        m = md if m is None else m & md
    return m


def _chunks(n, per_sample):
    step = max(1, _CHUNK_ELEMS // max(1, per_sample))
    for a in range(0, n, step):
        yield a, min(n, a + step)


def _prep(grid, displacement):
    if hasattr(grid, 'dense'):
        grid = grid.dense()
    dim = grid.shape[-1]
    coords = grid.reshape(grid.shape[0], -1, dim)
    if displacement:
        from .api import identity_grid
        ident = identity_grid(grid.shape[1:-1], dtype=grid.dtype, device=grid.device)
        coords = coords + ident.reshape(1, -1, dim)
    return coords, list(grid.shape[1:-1])


def _gather_op(inp, grid, bound, order, extrapolate, displacement, need):
    """pull (need 0) -> (B,C,*out); grad (need 1) -> (B,C,*out,D); hess (need 2) -> (B,C,*out,D,D)."""
    coords, oshape = _prep(grid, displacement)
    dim = coords.shape[-1]
    ishape = list(inp.shape[2:])
    B = max(inp.shape[0], coords.shape[0])
    C = inp.shape[1]
    dtype = torch.promote_types(inp.dtype, coords.dtype)
    flat = inp.reshape(inp.shape[0], C, -1).to(dtype).expand(B, C, -1)
    coords = coords.to(dtype).expand(B, -1, dim)
    N = coords.shape[1]
    trail = [] if need == 0 else ([dim] if need == 1 else [dim, dim])
    out = flat.new_zeros([B, C, N] + trail)
    ntap = 1
    for k in order:
        ntap *= k + 1
    for a, b in _chunks(N, B * C * ntap):
        cc = coords[:, a:b]
        st = _Stencil(cc, ishape, bound, order, need)
        lin = st.linear_index(ishape)                                            # (B, n, T)
        vals = torch.gather(flat, 2, lin.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, b - a, -1)
        m = _mask(cc, ishape, extrapolate)
        mf = None if m is None else m.to(dtype)[:, None]
        if need == 0:
            r = (vals * st.weights()[:, None]).sum(-1)
            out[:, :, a:b] = r if mf is None else r * mf
        elif need == 1:
            for d in range(dim):
                r = (vals * st.weights((d,))[:, None]).sum(-1)
                out[:, :, a:b, d] = r if mf is None else r * mf
        else:
            for d in range(dim):
                for e in range(d, dim):
                    r = (vals * st.weights((d, e))[:, None]).sum(-1)
                    r = r if mf is None else r * mf
                    out[:, :, a:b, d, e] = r
                    if e != d:
                        out[:, :, a:b, e, d] = r
    return out.reshape([B, C] + oshape + trail)


def _scatter_op(inp, grid, shape, bound, order, extrapolate, displacement, trailing, out=None):
    """push (trailing 0: inp (B,C,*in)) / pushgrad (trailing 1: inp (B,C,*in,D)) / count (inp None) -> (B,C,*shape).
    `out` (1 or B, C, *shape): accumulate into it (a batch of 1 is shared by all items)."""
    coords, sshape = _prep(grid, displacement)
    dim = coords.shape[-1]
    shape = sshape if shape is None else [int(s) for s in shape]
    B = coords.shape[0] if inp is None else max(inp.shape[0], coords.shape[0])
    dtype = coords.dtype if inp is None else torch.promote_types(inp.dtype, coords.dtype)
    coords = coords.to(dtype).expand(B, -1, dim)
    N = coords.shape[1]
    if inp is None:
        C = 1
        src = None
    else:
        C = inp.shape[1]
        src = inp.to(dtype).reshape([inp.shape[0], C, N] + ([dim] if trailing else [])).expand([B, C, N] + ([dim] if trailing else []))
    nvox = 1
    for n in shape:
        nvox *= n
    if out is None:
        acc = coords.new_zeros([B, C, nvox])
    else:
        acc = out.reshape(out.shape[0], C, nvox)
    ntap = 1
    for k in order:
        ntap *= k + 1
    for a, b in _chunks(N, B * C * ntap):
        cc = coords[:, a:b]
        st = _Stencil(cc, shape, bound, order, 1 if trailing else 0)
        lin = st.linear_index(shape)                                             # (B, n, T)
        m = _mask(cc, shape, extrapolate)
        mf = None if m is None else m.to(dtype)[:, None]
        if trailing:
            contrib = None
            for d in range(dim):
                s = src[:, :, a:b, d]
                s = s if mf is None else s * mf                                  # nd.py:346-347: masked before the scatter
                term = s[..., None] * st.weights((d,))[:, None]
                contrib = term if contrib is None else contrib + term
        else:
            w = st.weights()[:, None]
            if src is None:
                contrib = w if mf is None else w * mf[..., None]
            else:
                s = src[:, :, a:b]
                s = s if mf is None else s * mf                                  # nd.py:201-203
                contrib = s[..., None] * w
        contrib = contrib.expand(B, C, b - a, ntap).reshape(B, C, -1)
        index = lin.reshape(B, 1, -1).expand(B, C, -1)
        if acc.shape[0] == 1 and B > 1:                                          # shared target: every item adds into it
            acc[0].scatter_add_(1, index.permute(1, 0, 2).reshape(C, -1), contrib.permute(1, 0, 2).reshape(C, -1))
        else:
            acc.scatter_add_(2, index, contrib)
    if out is not None:
        return out
    return acc.reshape([B, C] + shape)


class TorchKernels:
    """Same interface as `ops._HipKernels`; any device, any number of spatial dims."""

    @staticmethod
    def pull(inp, grid, bound, order, extrapolate, displacement=False):
        return _gather_op(inp, grid, bound, order, extrapolate, displacement, 0)

    @staticmethod
    def grad(inp, grid, bound, order, extrapolate, displacement=False):
        return _gather_op(inp, grid, bound, order, extrapolate, displacement, 1)

    @staticmethod
    def hess(inp, grid, bound, order, extrapolate, displacement=False):
        return _gather_op(inp, grid, bound, order, extrapolate, displacement, 2)

    @staticmethod
    def push(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return _scatter_op(inp, grid, shape, bound, order, extrapolate, displacement, 0)

    @staticmethod
    def count(grid, shape, bound, order, extrapolate, displacement=False):
        return _scatter_op(None, grid, shape, bound, order, extrapolate, displacement, 0)

    @staticmethod
    def pushgrad(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return _scatter_op(inp, grid, shape, bound, order, extrapolate, displacement, 1)

    @staticmethod
    def push_count(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return torch.cat([TorchKernels.push(inp, grid, shape, bound, order, extrapolate, displacement),
                          TorchKernels.count(grid, shape, bound, order, extrapolate, displacement).to(inp.dtype)], 1)

    @staticmethod
    def push_shared_(out, inp, grid, bound, order, extrapolate, with_count=False):
        shape = list(out.shape[2:])
        if with_count:
            TorchKernels.push_shared_(out[:, :-1], inp, grid, bound, order, extrapolate)
            return TorchKernels.push_shared_(out[:, -1:], None, grid, bound, order, extrapolate)
        tmp = _scatter_op(inp, grid, shape, bound, order, extrapolate, False, 0)
        out += tmp.sum(0, keepdim=True).to(out.dtype)
        return out

    # backward compositions (pushpull.py:237-299)
    @staticmethod
    def pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        dim = grid.shape[-1]
        gi = gg = None
        if need_inp:
            gi = TorchKernels.push(grad, grid, list(inp.shape[-dim:]), bound, order, extrapolate, displacement)
            if inp.shape[0] == 1 and gi.shape[0] > 1:
                gi = gi.sum(0, keepdim=True)
        if need_grid:
            gg = (TorchKernels.grad(inp, grid, bound, order, extrapolate, displacement) * grad.unsqueeze(-1)).sum(1)
        return gi, gg

    @staticmethod
    def push_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        gi = gg = None
        if inp is None:                                               # count: pushpull.py:286-299
            return None, TorchKernels.grad(grad, grid, bound, order, extrapolate, displacement).sum(1)
        if need_inp:
            gi = TorchKernels.pull(grad, grid, bound, order, extrapolate, displacement)
            if inp.shape[0] == 1 and gi.shape[0] > 1:
                gi = gi.sum(0, keepdim=True)
        if need_grid:
            gg = (TorchKernels.grad(grad, grid, bound, order, extrapolate, displacement) * inp.unsqueeze(-1)).sum(1)
        return gi, gg

    @staticmethod
    def count_backward(grad, grid, bound, order, extrapolate, displacement=False):
        return TorchKernels.push_backward(grad, None, grid, bound, order, extrapolate, False, True, displacement)[1]

    # prefilter (coeff.py:258-284), one dim, in place
    @staticmethod
    def spline_filter_(data, bound, order, dim, src=None):
        from .filter_torch import spline_filter_
        return spline_filter_(data, bound, order, dim, src=src)

    @staticmethod
    def pull_labels(inp, grid, bound, order, extrapolate, displacement=False):
        raise NotImplementedError('the fused label-map kernel exists on the GPU only; interpol.api loops over the labels instead')

    @staticmethod
    def resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint, n_lattice):
        raise NotImplementedError('separable resampling passes exist on the GPU only; interpol.resize / restrict use a dense grid instead')
