#!/usr/bin/env python
"""Owner-computes push at config 2: own_plan + own_taps (default) against own_accumulate alone (debug bit 2048): time and equality."""
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
OLD = 2048 << 8
for sigma in [float(s) for s in sys.argv[1:]] or [2.0, 0.0, 6.0]:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    res = {"sigma": sigma}
    for name, kind, src in (("push", "push", inp), ("count", "count", None)):
        a = _hip.scatter(kind, src, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
        b = _hip.scatter(kind, src, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | OLD)
        res[name + "_ndiff"] = int((a != b).sum()); res[name + "_maxdiff"] = float((a - b).abs().max()); res[name + "_max"] = float(b.abs().max())
        res[name + "_new"] = round(timeit(lambda: _hip.scatter(kind, src, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)), 4)
        res[name + "_old"] = round(timeit(lambda: _hip.scatter(kind, src, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | OLD)), 4)
    print(json.dumps(res), flush=True)
