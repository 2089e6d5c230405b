#!/usr/bin/env python
"""Trilinear 3-D pull / grad / push at the config-2 shape: routed kernel, forced tiles, generic kernel, per sigma."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=4):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort(); return ts[len(ts) // 2]
for sigma in [float(s) for s in sys.argv[1:]] or [0.0, 0.5, 1.0, 2.0]:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    res = {}
    for op in ("pull", "grad"):
        for name, fl in (("routed", 0), ("tiles", _hip.FLAG_FORCE_TILED), ("generic", _hip.FLAG_NO_FASTPATH)):
            res[op + "_" + name] = round(timeit(lambda: _hip.gather(op, inp, grid, [3] * 3, [1] * 3, 1, flags=fl)), 3)
    for name, fl in (("routed", 0), ("tiles", _hip.FLAG_FORCE_TILED), ("generic", _hip.FLAG_NO_FASTPATH)):
        res["push_" + name] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [1] * 3, 1, flags=fl)), 3)
    print("sigma", sigma, json.dumps(res), flush=True)
