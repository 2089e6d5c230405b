#!/usr/bin/env python
"""pull_sorted: persistent workgroups against 1 / 2 / 4 tiles per workgroup (debug bits 13-15).  usage: tools/ab_pull_grid.py"""
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
for sigma in [2.0, 0.0, 4.0]:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    for _ in range(10): _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)
    res = {"persistent": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)), 3)}
    for m in (1, 2, 4):
        res["tiles_per_wg_%d" % m] = round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=(m << 13) << 8)), 3)
    res["persistent_again"] = round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)), 3)
    print("sigma", sigma, json.dumps(res))
