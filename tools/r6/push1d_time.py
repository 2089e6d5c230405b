#!/usr/bin/env python
"""1-D push / count / pull at the API sweep's shape (64 x 4 x 65536 float32, dct2, identity + N(0, sigma^2)): ms per call of the default routing
(round 6: csrc/push1d.hip) and of the generic kernel (INTERPOL_FLAG_NO_FASTPATH, the routing of rounds 1-5).  argv: [sigma]."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch
from interpol import _hip
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
B, C, n = 64, 4, 65536
g = torch.Generator().manual_seed(3)
src = torch.randn([B, C, n], generator=g).to(dev)
grid = (torch.arange(n, dtype=torch.float32) + sigma * torch.randn([B, n], generator=g))[..., None].contiguous().to(dev)

def timeit(fn, reps=9, inner=5):
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
    return round(ts[len(ts) // 2], 4)

for order in (1, 3, 5, 7):
    row = {"order": order, "sigma": sigma}
    for name, fl in (("tiles", 0), ("generic", _hip.FLAG_NO_FASTPATH)):
        row["push_" + name] = timeit(lambda: _hip.scatter("push", src, grid, [n], [3], [order], 1, flags=fl))
        row["count_" + name] = timeit(lambda: _hip.scatter("count", None, grid, [n], [3], [order], 1, flags=fl))
    row["push_bf16_tiles"] = timeit(lambda: _hip.scatter("push", src.bfloat16(), grid, [n], [3], [order], 1))
    row["pull"] = timeit(lambda: _hip.gather("pull", src, grid, [3], [order], 1))
    row["pull_backward_both"] = timeit(lambda: _hip.pull_backward(src, src, grid, [3], [order], 1, True, True))
    print(json.dumps(row), flush=True)
