#!/usr/bin/env python
"""Orders 4 and 5 (BASELINE config 3's shape: 8 x 1 x 192^3, dft): grid_push through bricks of the target (scatter5, the default with a
workspace) against the LDS tiles (backend.rough_deformations = False) and the generic kernel; the pull's backward with both gradients."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
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
B, C, n = 8, 1, 192
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
ident = interpol.identity_grid([n, n, n], device=dev)[None]
bc = [4] * 3     # dft
for order in [int(a) for a in (os.environ.get("ORDERS", "5,4").split(","))]:
    for sigma in [float(a) for a in sys.argv[1:]] or [0.0, 2.0, 6.0]:
        grid = (ident + sigma * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
        o = [order] * 3
        res = {"order": order, "sigma": sigma}
        backend.rough_deformations = None
        a = _hip.scatter("push", x, grid, None, bc, o, 1)
        res["push_bricks"] = round(timeit(lambda: _hip.scatter("push", x, grid, None, bc, o, 1)), 3)
        res["push_bricks_forced"] = round(timeit(lambda: _hip.scatter("push", x, grid, None, bc, o, 1, flags=_hip.FLAG_BINNED_SCATTER)), 3)
        res["bwd_both_bricks"] = round(timeit(lambda: _hip.pull_backward(x, x, grid, bc, o, 1, True, True)), 3)
        ga = _hip.pull_backward(x, x, grid, bc, o, 1, True, True)
        backend.rough_deformations = False
        if sigma <= 3:
            res["push_tiles"] = round(timeit(lambda: _hip.scatter("push", x, grid, None, bc, o, 1)), 3)
            res["bwd_both_tiles"] = round(timeit(lambda: _hip.pull_backward(x, x, grid, bc, o, 1, True, True)), 3)
        backend.rough_deformations = None
        r = _hip.scatter("push", x, grid, None, bc, o, 1, flags=_hip.FLAG_NO_FASTPATH)
        res["push_rel_err"] = "%.1e" % float((a - r).abs().max() / r.abs().max())
        gr_ = _hip.pull_backward(x, x, grid, bc, o, 1, True, True, flags=_hip.FLAG_NO_FASTPATH)
        res["bwd_rel_err"] = "%.1e" % max(float((u - v).abs().max() / v.abs().max()) for u, v in zip(ga, gr_))
        print(json.dumps(res), flush=True)
        del a, r, ga, gr_, grid
