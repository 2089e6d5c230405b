#!/usr/bin/env python
"""sweep_many_tiles.py's regime (several tiles per workgroup) for the other forms of input: mixed orders, displacement fields, separable lattices,
a target shared by the batch items."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
from interpol.sepgrid import SeparableGrid
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(12)
NF = _hip.FLAG_NO_FASTPATH
bad = 0
def check(name, got, ref, tol, what):
    global bad
    got = got if isinstance(got, (tuple, list)) else [got]
    ref = ref if isinstance(ref, (tuple, list)) else [ref]
    for i, (a, r) in enumerate(zip(got, ref)):
        if a is None:
            continue
        e = float((a.float() - r.float()).abs().max() / max(float(r.float().abs().max()), 1e-30))
        if not e < tol:
            bad += 1
            print("BAD", name, i, what, e, flush=True)
for dim, shape in ((3, (112, 96, 104)), (2, (1500, 1100))):
    ident = interpol.identity_grid(shape)[None].to(dev)
    for orders in (([1, 2, 3][:dim], [3, 1, 2][:dim]) if os.environ.get("SWEEP_TRIM") == "1" else ([1, 2, 3][:dim], [3, 1, 2][:dim], [0, 3, 3][:dim], [1] * dim, [3] * dim)):
        for C in (1, 2):
            for sigma in (0.3, 4.0):
                vol = torch.randn([2, C, *shape], generator=gen).to(dev)
                src = torch.randn([2, C, *shape], generator=gen).to(dev)
                disp = (sigma * torch.randn([2, *shape, dim], generator=gen)).to(dev)
                grid = (ident + disp).contiguous()
                b = [3, 1, 6][:dim]
                what = (dim, orders, C, sigma)
                for gname, gr, fl in (("dense", grid, 0), ("disp", disp, _hip.FLAG_DISPLACEMENT)):
                    w = what + (gname,)
                    check("pull", _hip.gather("pull", vol, gr, b, orders, 1, flags=fl), _hip.gather("pull", vol, gr, b, orders, 1, flags=fl | NF), 1e-5, w)
                    check("grad", _hip.gather("grad", vol, gr, b, orders, 1, flags=fl), _hip.gather("grad", vol, gr, b, orders, 1, flags=fl | NF), 2e-5, w)
                    check("push", _hip.scatter("push", src, gr, list(shape), b, orders, 1, flags=fl, with_count=True), _hip.scatter("push", src, gr, list(shape), b, orders, 1, flags=fl | NF, with_count=True), 1e-5, w)
                    for nv, ng in ((True, True), (True, False), (False, True)):
                        check("pull_backward %d%d" % (nv, ng), _hip.pull_backward(src, vol, gr, b, orders, 1, nv, ng, flags=fl), _hip.pull_backward(src, vol, gr, b, orders, 1, nv, ng, flags=fl | NF), 2e-5, w)
                        check("push_backward %d%d" % (nv, ng), _hip.push_backward(vol, src, gr, b, orders, 1, nv, ng, flags=fl), _hip.push_backward(vol, src, gr, b, orders, 1, nv, ng, flags=fl | NF), 2e-5, w)
                # a separable lattice (zoom 0.9 + offset), against the dense grid it stands for
                if sigma == 0.3:
                    lins = [torch.linspace(-1.5, n - 0.2, n).to(dev) for n in shape]
                    sep = SeparableGrid(lins)
                    dense = torch.stack(torch.meshgrid(*lins, indexing="ij"), -1)[None].expand(2, *shape, dim).contiguous()
                    w = what + ("separable",)
                    check("pull", _hip.gather("pull", vol, sep, b, orders, 1), _hip.gather("pull", vol, dense, b, orders, 1, flags=NF), 1e-5, w)
                    check("grad", _hip.gather("grad", vol, sep, b, orders, 1), _hip.gather("grad", vol, dense, b, orders, 1, flags=NF), 2e-5, w)
                    check("push", _hip.scatter("push", src, sep, list(shape), b, orders, 1), _hip.scatter("push", src, dense, list(shape), b, orders, 1, flags=NF), 1e-5, w)
                    check("pull_backward vol", _hip.pull_backward(src, vol, sep, b, orders, 1, True, False)[0], _hip.pull_backward(src, vol, dense, b, orders, 1, True, False, flags=NF)[0], 1e-5, w)
                # ONE target shared by the batch items
                if dim == 3:
                    ref = _hip.scatter("push", src, grid, list(shape), b, orders, 1, flags=NF, with_count=True).sum(0, keepdim=True)
                    got = _hip.scatter("push", src, grid, list(shape), b, orders, 1, shared=True, with_count=True)
                    check("shared push", got, ref, 1e-5, what)
                torch.cuda.synchronize()
        print("done", dim, orders, "bad so far", bad, flush=True)
print("sweep2: bad =", bad, flush=True)
sys.exit(1 if bad else 0)
