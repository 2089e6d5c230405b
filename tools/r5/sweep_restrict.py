#!/usr/bin/env python
"""restrict / the backward of resize at benchmark sizes: the adjoint gathering passes (csrc/resample1d.hip: resample1d_adj_gather) against the
ONE D-dimensional push on the separable lattice they replaced (interpol/separable.py with the gathers switched off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import separable
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(15)
bad = 0
orig = separable._gathers
for shape, factor in (((256, 256, 256), 2), ((200, 180, 220), 3), ((2000, 1500), 2), ((300000,), 4)):
    D = len(shape)
    for order in (1, 2, 3):
        for bound in ("dct2", "zero", "dft"):
            for anchor in ("e", "c", "f"):
                x = torch.randn([2, 2, *shape], generator=gen).to(dev)
                kw = dict(factor=[factor] * D, anchor=anchor, interpolation=order, bound=bound)
                got = interpol.restrict(x, **kw)
                separable._gathers = lambda *a: False
                try:
                    ref = interpol.restrict(x, **kw)
                finally:
                    separable._gathers = orig
                e = float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))
                # and the backward of resize (autograd): the same adjoint passes
                y = torch.randn([1, 1, *[s // factor for s in shape]], generator=gen).to(dev).requires_grad_()
                up = interpol.resize(y, factor=[factor] * D, anchor=anchor, interpolation=order, bound=bound)
                w = torch.randn(up.shape, generator=gen).to(dev)
                g1, = torch.autograd.grad((up * w).sum(), y)
                separable._gathers = lambda *a: False
                try:
                    up2 = interpol.resize(y, factor=[factor] * D, anchor=anchor, interpolation=order, bound=bound)
                    g2, = torch.autograd.grad((up2 * w).sum(), y)
                finally:
                    separable._gathers = orig
                e2 = float((g1 - g2).abs().max() / max(float(g2.abs().max()), 1e-30))
                if not (e < 2e-5 and e2 < 2e-5):
                    bad += 1
                    print("BAD", shape, factor, order, bound, anchor, e, e2, flush=True)
    print("done", shape, "bad so far", bad, flush=True)
print("sweep restrict: bad =", bad, flush=True)
