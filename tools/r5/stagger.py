#!/usr/bin/env python
"""(experiment, -DIP_STAGGER build) owner-computes push at config 2 with the second workgroup of each CU started late:
dbg bits 9-11 = delay in units of 8192 clocks; bit 12: 'late' = second half of the grid instead of odd TG_ID."""
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
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
res = {}
for half in (0, 8):
    for d in range(8):
        dbg = (d << 9) | (half << 9)
        res["%s%d" % ("h" if half else "t", d)] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | (dbg << 8))), 4)
print("sigma", sigma, json.dumps(res), flush=True)
