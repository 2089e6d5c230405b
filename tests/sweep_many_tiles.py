#!/usr/bin/env python
"""Default routing against the atomics-only / generic kernels at sizes where a workgroup serves several tiles (the regime the unit tests'
small shapes do not reach): every operator, dims 2 - 3, orders 0 - 7, one and three channels, smooth and rough fields."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(11)
NF = _hip.FLAG_NO_FASTPATH
bad = 0
def err(a, r):
    return float((a.float() - r.float()).abs().max() / max(float(r.float().abs().max()), 1e-30))
def check(name, got, ref, tol, what):
    global bad
    got = got if isinstance(got, (tuple, list)) else [got]
    ref = ref if isinstance(ref, (tuple, list)) else [ref]
    for i, (a, r) in enumerate(zip(got, ref)):
        if a is None:
            continue
        e = err(a, r)
        if not e < tol:
            bad += 1
            print("BAD", name, i, what, e, flush=True)
BIG = os.environ.get("SWEEP_BIG") == "1"          # 2 x C x 256^3 / 2 x C x 4096^2, orders 1 and 3: the sizes of the benchmark configs
MANY = os.environ.get("SWEEP_MANY") == "1"        # many small batch items instead: 300 x C x 24x20x28 / 300 x C x 70x90
TRIM = os.environ.get("SWEEP_TRIM") == "1"        # the slice tests/test_fuzz_slices.py runs: orders 1, 3, 5, one bound per case
NB = 300 if MANY else 2
for dim, shape in (((3, (256, 256, 256)), (2, (4096, 4096))) if BIG else (((3, (24, 20, 28)), (2, (70, 90))) if MANY else ((3, (112, 96, 104)), (2, (1500, 1100))))):
    for order in ((1, 3) if BIG else ((1, 3, 5) if TRIM else (0, 1, 2, 3, 4, 5, 6, 7))):
        shape_ = shape                               # (round 6: orders 6 - 7 at the full shape too -- they go through bricks now)
        ident = interpol.identity_grid(shape_)[None]
        for C in (1, 3):
            for sigma in ((0.3, 6.0) if BIG else (0.3, 4.0)):
                for bound, ex in (((3, 1), (0, 0)) if BIG else ((((3, 1), (0, 0), (6, 2))[(order + C) % 3],) if TRIM else ((3, 1), (0, 0), (6, 2)))):
                    vol = torch.randn([NB, C, *shape_], generator=gen).to(dev)
                    src = torch.randn([NB, C, *shape_], generator=gen).to(dev)
                    grid = (ident + sigma * torch.randn([NB, *shape_, dim], generator=gen)).contiguous().to(dev)
                    b, o = [bound] * dim, [order] * dim
                    what = (dim, order, C, sigma, bound, ex)
                    check("pull", _hip.gather("pull", vol, grid, b, o, ex), _hip.gather("pull", vol, grid, b, o, ex, flags=NF), 1e-5, what)
                    check("grad", _hip.gather("grad", vol, grid, b, o, ex), _hip.gather("grad", vol, grid, b, o, ex, flags=NF), 2e-5, what)
                    check("push", _hip.scatter("push", src, grid, list(shape_), b, o, ex, with_count=True), _hip.scatter("push", src, grid, list(shape_), b, o, ex, flags=NF, with_count=True), 1e-5, what)
                    check("count", _hip.scatter("count", None, grid, list(shape_), b, o, ex), _hip.scatter("count", None, grid, list(shape_), b, o, ex, flags=NF), 1e-5, what)
                    for nv, ng in ((True, True), (True, False), (False, True)):
                        check("pull_backward %d%d" % (nv, ng), _hip.pull_backward(src, vol, grid, b, o, ex, nv, ng), _hip.pull_backward(src, vol, grid, b, o, ex, nv, ng, flags=NF), 2e-5, what)
                        check("push_backward %d%d" % (nv, ng), _hip.push_backward(vol, src, grid, b, o, ex, nv, ng), _hip.push_backward(vol, src, grid, b, o, ex, nv, ng, flags=NF), 2e-5, what)
                    if bound == 3:
                        for dt, tol in ((torch.bfloat16, 2e-2),):
                            check("pull bf16", _hip.gather("pull", vol.to(dt), grid, b, o, ex), _hip.gather("pull", vol.to(dt).float(), grid, b, o, ex, flags=NF), tol, what)
                            check("push bf16", _hip.scatter("push", src.to(dt), grid, list(shape_), b, o, ex), _hip.scatter("push", src.to(dt).float(), grid, list(shape_), b, o, ex, flags=NF), tol, what)
                    torch.cuda.synchronize()
        print("done", dim, order, "bad so far", bad, flush=True)
print("sweep: bad =", bad, flush=True)
sys.exit(1 if bad else 0)
