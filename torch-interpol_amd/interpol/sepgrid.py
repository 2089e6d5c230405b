"""Separable (tensor-product) sampling lattices.

`resize` and `restrict` sample on `stack(meshgrid_ij(*lin), -1)` (reference
interpol/resize.py:96-123, restrict.py:88-117): the coordinate of sample
(o_0, .., o_{D-1}) along dim d is `lin[d][o_d]`.  A `SeparableGrid` carries the
D coordinate vectors instead of the (*out, D) tensor; the kernels read
`lin[d][o_d]` directly (INTERPOL_FLAG_SEPARABLE_GRID of include/interpol_hip.h):
no (*out, D) grid is written or read.  The coordinate VALUES are the ones the
reference computes (same torch ops on the same dtype), so results are those of
the dense-grid path bit for bit.

It quacks like a constant grid tensor of shape (1, *out, D) for the host code
(`shape`, `dtype`, `device`, `requires_grad = False`).
"""
import torch

__all__ = ['SeparableGrid', 'AffineGrid', 'LazyGrid']


class LazyGrid:
    """A constant sampling lattice given by a rule instead of a (*out, D) tensor: the kernels evaluate
    the rule in registers.  Quacks like a grid tensor of shape (1, *out, D) without gradient."""
    requires_grad = False


class SeparableGrid(LazyGrid):

    def __init__(self, lins):
        lins = [l.detach() for l in lins]
        if not 1 <= len(lins) <= 3 or any(l.dim() != 1 for l in lins):
            raise ValueError('SeparableGrid: expected 1 to 3 one-dimensional coordinate vectors')
        self.lins = lins

    @property
    def shape(self):
        return torch.Size([1, *[len(l) for l in self.lins], len(self.lins)])

    @property
    def dtype(self):
        return self.lins[0].dtype

    @property
    def device(self):
        return self.lins[0].device

    @property
    def is_cuda(self):
        return self.lins[0].is_cuda

    def dim(self):
        return len(self.lins) + 2

    def numel(self):
        n = len(self.lins)
        for l in self.lins:
            n *= len(l)
        return n

    def new_zeros(self, *a, **k):
        return self.lins[0].new_zeros(*a, **k)

    def to(self, *a, **k):
        return SeparableGrid([l.to(*a, **k) for l in self.lins])

    def packed(self, dtype):
        """lin_0 | lin_1 | lin_2 back to back, the layout the kernels index."""
        return torch.cat([l.to(dtype) for l in self.lins]).contiguous()

    def dense(self):
        """The (1, *out, D) tensor this lattice stands for."""
        return torch.stack(torch.meshgrid(*self.lins, indexing='ij'), dim=-1)[None]


class AffineGrid(LazyGrid):
    """The lattice of `affine_grid(mat, shape)` (reference interpol/api.py:534-572) without the
    (*shape, D) tensor: sample o has coordinates A o + t, evaluated inside the kernels from the
    D x (D+1) matrix (INTERPOL_FLAG_AFFINE_GRID of include/interpol_hip.h): 4 D bytes per sample point
    less to read.  ONE matrix (D[+1], D+1) -- a batch of matrices, or a matrix that needs a gradient,
    goes through the dense `affine_grid`.  The coordinates are computed as ((A_d0 o_0) + A_d1 o_1 + ...)
    + t_d with fused multiply-adds in the grid dtype: equal to `affine_grid`'s matmul up to the
    rounding of that sum (bit-identical whenever the products are exact)."""

    def __init__(self, mat, shape):
        mat = torch.as_tensor(mat).detach()
        shape = [int(n) for n in shape]
        dim = mat.shape[-1] - 1
        if mat.dim() != 2 or dim != len(shape) or mat.shape[0] not in (dim, dim + 1) or not 1 <= dim <= 3:
            raise ValueError('AffineGrid: expected one (D[+1], D+1) matrix and a shape of length D <= 3')
        self.mat = mat[:dim]
        self._shape = shape

    @property
    def shape(self):
        return torch.Size([1, *self._shape, len(self._shape)])

    @property
    def dtype(self):
        return self.mat.dtype

    @property
    def device(self):
        return self.mat.device

    @property
    def is_cuda(self):
        return self.mat.is_cuda

    def dim(self):
        return len(self._shape) + 2

    def numel(self):
        n = len(self._shape)
        for m in self._shape:
            n *= m
        return n

    def new_zeros(self, *a, **k):
        return self.mat.new_zeros(*a, **k)

    def to(self, *a, **k):
        return AffineGrid(self.mat.to(*a, **k), self._shape)

    def packed(self, dtype):
        """the D x (D+1) matrix [A | t], row-major: what the kernels read"""
        return self.mat.to(dtype).contiguous().reshape(-1)

    def dense(self):
        """The (1, *shape, D) tensor this lattice stands for (same operation order as the kernels, unfused)."""
        dim = len(self._shape)
        o = torch.stack(torch.meshgrid(*[torch.arange(n, dtype=self.mat.dtype, device=self.mat.device) for n in self._shape],
                                       indexing='ij'), -1)
        cols = []
        for d in range(dim):
            s = self.mat[d, 0] * o[..., 0]
            for e in range(1, dim):
                s = s + self.mat[d, e] * o[..., e]               # (the kernels fuse these multiply-adds)
            cols.append(s + self.mat[d, dim])
        return torch.stack(cols, -1)[None]
