#!/usr/bin/env python
"""Push (FLAG_BINNED_SCATTER) at config 2 under a list of debug-bit sets: ms per call.  argv: sigma, then the bit sets (ints, the
library's KParams::dbg)."""
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
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
sets = [int(a) for a in sys.argv[2:]] or [0]
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
out = torch.zeros_like(inp)
ref = None
res = {}
for bits in sets:
    fl = _hip.FLAG_BINNED_SCATTER | (bits << 8)
    r = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=fl)
    if ref is None:
        ref = r
    d = (r - ref).abs().max().item() / ref.abs().max().item()
    del r
    # target allocated outside: the op alone (accumulates into `out`)
    t = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=fl | _hip.FLAG_ACCUMULATE, out=out))
    res[str(bits)] = {"ms": round(t, 4), "diff_vs_first": d}
print(json.dumps(res), flush=True)
