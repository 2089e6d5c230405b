#!/usr/bin/env python
"""float64 timings at 1x2x128^3 cubic / dct2 (the reference's own tests are float64): pull, push, count, grad, both backward
passes; sigma = 2 and the identity.  usage: tools/time_f64.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, batch=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort(); return ts[len(ts) // 2]
n = 128
for sigma in (2.0, 0.0):
    g = torch.Generator().manual_seed(3)
    inp = torch.randn([1, 2, n, n, n], generator=g, dtype=torch.float64).to(dev)
    grid = (interpol.identity_grid([n] * 3, dtype=torch.float64)[None] + sigma * torch.randn([1, n, n, n, 3], generator=g, dtype=torch.float64)).to(dev)
    kw = dict(interpolation=3, bound="dct2", extrapolate=True)
    res = {"pull": timeit(lambda: interpol.grid_pull(inp, grid, **kw)), "push": timeit(lambda: interpol.grid_push(inp, grid, **kw)),
           "count": timeit(lambda: interpol.grid_count(grid, **kw)), "grad": timeit(lambda: interpol.grid_grad(inp, grid, **kw))}
    gi = inp.clone().requires_grad_(True); gg = grid.clone().requires_grad_(True)
    def bwd():
        out = interpol.grid_pull(gi, gg, **kw); out.backward(torch.ones_like(out)); gi.grad = None; gg.grad = None
    res["pull_fwd_bwd"] = timeit(bwd)
    print("f64 1x2x128^3 cubic dct2 sigma", sigma, json.dumps({k: round(v, 3) for k, v in res.items()}))
