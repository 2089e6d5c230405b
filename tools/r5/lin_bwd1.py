#!/usr/bin/env python
"""Backward of the pull with ONE channel and both gradients (orders 0, 1): the library's fused kernel against the split push + grid gradient."""
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
B, C, n = 4, 1, 256
ident = interpol.identity_grid([n, n, n], device=dev)[None]
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
for s in (0.0, 0.5, 2.0, 6.0):
    grid = (ident + s * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
    res = {"sigma": s}
    for o in (1, 3):
        res["o%d_fused" % o] = round(timeit(lambda: _hip.pull_backward(x, x, grid, [3] * 3, [o] * 3, 1, True, True)), 3)
        res["o%d_push" % o] = round(timeit(lambda: _hip.scatter("push", x, grid, [n] * 3, [3] * 3, [o] * 3, 1)), 3)
        res["o%d_gridonly" % o] = round(timeit(lambda: _hip.pull_backward(x, x, grid, [3] * 3, [o] * 3, 1, False, True)), 3)
    print(json.dumps(res), flush=True)
    del grid
