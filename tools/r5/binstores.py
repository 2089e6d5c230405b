#!/usr/bin/env python
"""(experiment, -DIP_STAGGER build) own_bin with its scattered stores switched off: dbg 64 no `meta` store, 128 no `vals` store (wrong results: timing only)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=7, inner=4):
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
inp, grid = bench.make_inputs(4, 2, 256, 2.0, dev, 1234)
res = {}
for dbg in (0, 64, 128, 192, 0):
    res[str(dbg)] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | (dbg << 8))), 4)
print(json.dumps(res), flush=True)
