#!/usr/bin/env python
"""Config 2: unrouted pull (sample tiles), routed pull, routed push, owner push -- for A/B runs of library variants."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
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
res = {"lib": os.path.basename(os.environ.get("INTERPOL_HIP_LIB", "default"))}
for sigma in [float(s) for s in sys.argv[1:]] or [2.0, 0.0]:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    res["pull_tiles_%g" % sigma] = round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_FORCE_TILED)), 4)
    res["pull_%g" % sigma] = round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)), 4)
    res["push_%g" % sigma] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1)), 4)
    res["push_owner_%g" % sigma] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)), 4)
print(json.dumps(res), flush=True)
