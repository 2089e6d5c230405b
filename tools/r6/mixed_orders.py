#!/usr/bin/env python
"""Mixed per-dim orders at config 2's shape (4 x 2 x 256^3 fp32, dct2, identity + N(0, sigma^2)): ms per call of pull, the grid gradient
of pull_backward and push_backward (both gradients) through the default routing -- round 6: the class-sorted cubic tiles with runtime
per-dim weights -- and with debug bit 16 (the round-1 tiles of ops_tiled.hip, the routing of rounds 1-5).  argv: [sigma]."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, bench
from interpol import _hip
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
src = torch.randn_like(inp)

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
    return round(ts[len(ts) // 2], 3)

b = [3, 3, 3]
for orders in ([1, 2, 3], [3, 1, 2], [2, 2, 3], [3, 3, 3], [2, 2, 2]):
    row = {"orders": orders, "sigma": sigma}
    for name, fl in (("sorted", 0), ("round1_tiles", 16 << 8)):
        row["pull_" + name] = timeit(lambda: _hip.gather("pull", inp, grid, b, orders, 1, flags=fl))
        row["ggrid_" + name] = timeit(lambda: _hip.pull_backward(src, inp, grid, b, orders, 1, False, True, flags=fl))
        row["pushbwd_" + name] = timeit(lambda: _hip.push_backward(inp, src, grid, b, orders, 1, True, True, flags=fl))
    row["push"] = timeit(lambda: _hip.scatter("push", src, grid, [256] * 3, b, orders, 1))
    print(json.dumps(row), flush=True)
