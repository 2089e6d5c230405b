#!/usr/bin/env python
"""sweep_many_tiles.py's regime in float64 and float16 storage (3-D and 2-D, orders 1 - 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(13)
NF = _hip.FLAG_NO_FASTPATH
bad = 0
def check(name, got, ref, tol, what):
    global bad
    got = got if isinstance(got, (tuple, list)) else [got]
    ref = ref if isinstance(ref, (tuple, list)) else [ref]
    for i, (a, r) in enumerate(zip(got, ref)):
        if a is None:
            continue
        e = float((a.double() - r.double()).abs().max() / max(float(r.double().abs().max()), 1e-30))
        if not e < tol:
            bad += 1
            print("BAD", name, i, what, e, flush=True)
for dim, shape in ((3, (112, 96, 104)), (2, (1500, 1100))):
    ident = interpol.identity_grid(shape)[None]
    for dt, gdt, tol in ((torch.float64, torch.float64, 1e-11), (torch.float16, torch.float32, 4e-3)):
        for order in (1, 2, 3):
            for C in (1, 2):
                for sigma in (0.3, 4.0):
                    vol = torch.randn([2, C, *shape], generator=gen).to(dt).to(dev)
                    src = torch.randn([2, C, *shape], generator=gen).to(dt).to(dev)
                    grid = (ident + sigma * torch.randn([2, *shape, dim], generator=gen)).to(gdt).contiguous().to(dev)
                    b, o = [3, 0, 6][:dim], [order] * dim
                    what = (dim, str(dt), order, C, sigma)
                    up = (lambda t: t) if dt == torch.float64 else (lambda t: t.float())
                    check("pull", _hip.gather("pull", vol, grid, b, o, 1), _hip.gather("pull", up(vol), grid, b, o, 1, flags=NF), tol, what)
                    check("grad", _hip.gather("grad", vol, grid, b, o, 1), _hip.gather("grad", up(vol), grid, b, o, 1, flags=NF), tol, what)
                    check("push", _hip.scatter("push", src, grid, list(shape), b, o, 1, with_count=True), _hip.scatter("push", up(src), grid, list(shape), b, o, 1, flags=NF, with_count=True), tol, what)
                    for nv, ng in ((True, True), (True, False), (False, True)):
                        check("pull_backward %d%d" % (nv, ng), _hip.pull_backward(src, vol, grid, b, o, 1, nv, ng), _hip.pull_backward(up(src), up(vol), grid, b, o, 1, nv, ng, flags=NF), tol * (4 if dt == torch.float16 else 1), what)
                        check("push_backward %d%d" % (nv, ng), _hip.push_backward(vol, src, grid, b, o, 1, nv, ng), _hip.push_backward(up(vol), up(src), grid, b, o, 1, nv, ng, flags=NF), tol * (4 if dt == torch.float16 else 1), what)
                    torch.cuda.synchronize()
        print("done", dim, dt, "bad so far", bad, flush=True)
print("sweep3: bad =", bad, flush=True)
sys.exit(1 if bad else 0)
