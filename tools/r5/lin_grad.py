#!/usr/bin/env python
"""Trilinear grid_grad: default against the generic kernel and the forced tiles, over the roughness of the field."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]
g = torch.Generator(device=dev).manual_seed(2)
B, C, n = 4, 2, 256
ident = interpol.identity_grid([n, n, n], device=dev)[None]
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
for s in (0.0, 0.5, 1.0, 2.0, 4.0):
    grid = (ident + s * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
    res = {"sigma": s}
    for o in (1, 3):
        for name, fl in (("default", 0), ("generic", _hip.FLAG_NO_FASTPATH), ("tiles", _hip.FLAG_FORCE_TILED)):
            res["grad_o%d_%s" % (o, name)] = round(timeit(lambda: _hip.gather("grad", x, grid, [3] * 3, [o] * 3, 1, flags=fl)), 3)
    print(json.dumps(res), flush=True)
    del grid
