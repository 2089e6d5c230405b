#!/usr/bin/env python
"""grid_pull (sample tiles alone: no workspace) at config 2 under sets of KParams::dbg bits: ms per call.  argv: sigma, bit sets."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=9, inner=4):
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
sigma = float(sys.argv[1])
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
res = {}
for bits in [int(a) for a in sys.argv[2:]]:
    res[str(bits)] = round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=(bits << 8))), 4)
print(json.dumps(res))
