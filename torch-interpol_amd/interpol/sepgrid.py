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

__all__ = ['SeparableGrid']


class SeparableGrid:
    requires_grad = False

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
