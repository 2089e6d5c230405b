#!/usr/bin/env python
"""Cubic grid_pull at config 2's shape on friendly fields: the generic kernel (no LDS, INTERPOL_FLAG_NO_FASTPATH) against the default
routing (pull_sorted), ms per call."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=7, inner=3):
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
inp, grid = bench.make_inputs(4, 2, 256, 0.0, dev, 1234)
fields = {"identity": grid, "smooth_amp2": bench.smooth_grid(4, 256, 2.0, dev, 7), "smooth_amp8": bench.smooth_grid(4, 256, 8.0, dev, 7)}
for s in (0.25, 0.5, 1.0):
    fields["iid_%g" % s] = bench.make_inputs(4, 2, 256, s, dev, 1234)[1]
for name, g in fields.items():
    res = {}
    for order in (3, 2):
        o = [order] * 3
        res["o%d_default" % order] = round(timeit(lambda: _hip.gather("pull", inp, g, [3] * 3, o, 1)), 3)
        res["o%d_generic" % order] = round(timeit(lambda: _hip.gather("pull", inp, g, [3] * 3, o, 1, flags=_hip.FLAG_NO_FASTPATH)), 3)
    print(name, json.dumps(res), flush=True)
