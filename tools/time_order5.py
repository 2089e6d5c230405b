#!/usr/bin/env python
"""BASELINE config 3's shape (8 x 1 x 192^3, order 5, dft): grid_pull and grid_grad through the LDS tiles (default before round 4)
and through bricks of the image (csrc/gather5.hip, INTERPOL_FLAG_BINNED_SCATTER), on i.i.d. noise of sigma voxels; error of the
bricks against the generic kernel.  ms per call, median of 5 x 4 back-to-back calls.  usage: time_order5.py [order] [sigma ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)


def timeit(fn, reps=5, inner=4):
    for _ in range(3):
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


order = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(3)
n = 192
inp = torch.randn([8, 1, n, n, n], generator=g).to(dev)
B, T = _hip.FLAG_BINNED_SCATTER, _hip.FLAG_FORCE_TILED
for sigma in [float(a) for a in sys.argv[2:]] or (2.0, 0.0, 4.0):
    grid = (interpol.identity_grid([n] * 3)[None] + sigma * torch.randn([8, n, n, n, 3], generator=g)).to(dev)
    res = {}
    for op in ("pull", "grad"):
        f = lambda fl=0: _hip.gather(op, inp, grid, [6] * 3, [order] * 3, 1, flags=fl)
        ref = f(_hip.FLAG_NO_FASTPATH)
        res[op + "_default"] = round(timeit(f), 3)
        res[op + "_tiles"] = round(timeit(lambda: f(T)), 3)
        res[op + "_bricks"] = round(timeit(lambda: f(B)), 3)
        res[op + "_bricks_err"] = "%.1e" % float((f(B) - ref).abs().max() / ref.abs().max())
        del ref
    print("order", order, "sigma", sigma, res, flush=True)
