#!/usr/bin/env python
"""Owner-computes push at config 2 with phases of own_accumulate switched off (an -DIP_ABLATE build of push_owner.hip:
INTERPOL_HIP_LIB=.../libinterpol_hip_abl.so).  dbg bits: 1 no flush, 2 no taps, 4 tap arithmetic without the LDS adds,
16 LDS adds without the arithmetic."""
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
bits = [int(x) for x in sys.argv[2:]] or [0, 1, 2, 3, 4, 16, 5, 17]
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
out = torch.empty_like(inp)
res = {}
for d in bits:
    res[str(d)] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | (d << 8))), 4)
print("sigma", sigma, json.dumps(res), flush=True)
