#!/usr/bin/env python
"""ms per grid_pull (config 2 shape) through _hip.gather, batches of 4 back-to-back calls; the first entry is measured again at the
end (the first measurements of a process run at ramping clocks).  usage: tools/time_pull.py [sigma ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench
dev = torch.device("cuda", 0)
def timeit(fn, reps=7, batch=4):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort(); return ts[len(ts) // 2]
sigmas = [float(a) for a in sys.argv[1:]] or [2.0, 0.0]
for sigma in sigmas:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    gout = torch.randn_like(inp)
    fns = {"pull": lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1),
           "pull_no_handback": lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=256 << 8),
           "pull_quadratic": lambda: _hip.gather("pull", inp, grid, [3] * 3, [2] * 3, 1),
           "pull_dft": lambda: _hip.gather("pull", inp, grid, [6] * 3, [3] * 3, 1),
           "pull_dst2": lambda: _hip.gather("pull", inp, grid, [5] * 3, [3] * 3, 1),
           "grid_gradient_of_pull": lambda: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, False, True)}
    for _ in range(20): fns["pull"]()
    res = {k: round(timeit(f), 3) for k, f in fns.items()}
    res["pull_again"] = round(timeit(fns["pull"]), 3)
    print("sigma", sigma, json.dumps(res))
