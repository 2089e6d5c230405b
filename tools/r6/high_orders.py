#!/usr/bin/env python
"""Orders 6 and 7 at config 2's shape (4 x 2 x 256^3 fp32, dct2, identity + N(0, sigma^2)): pull, grid_grad and the grid gradient of
pull_backward through the default routing (round 6: csrc/gather7.hip, bricks of the image) and with the workspace withheld
(backend.rough_deformations = False: the round-1 tiles of ops_tiled.hip, the routing of rounds 1 - 5).  argv: [sigma]."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1][0] != "-" else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
src = torch.randn_like(inp)

def timeit(fn, reps=5, inner=2):
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
for order in (5, 6, 7):
    o = [order] * 3
    row = {"order": order, "sigma": sigma}
    for name, rough in (("bricks", None), ("tiles", False)):
        backend.rough_deformations = rough                       # False: no workspace, the LDS tiles of ops_tiled.hip
        row["pull_" + name] = timeit(lambda: _hip.gather("pull", inp, grid, b, o, 1))
        row["grad_" + name] = timeit(lambda: _hip.gather("grad", inp, grid, b, o, 1))
        row["ggrid_" + name] = timeit(lambda: _hip.pull_backward(src, inp, grid, b, o, 1, False, True))
        if "--push" in sys.argv:
            row["push_" + name] = timeit(lambda: _hip.scatter("push", src, grid, [256] * 3, b, o, 1))
    backend.rough_deformations = None
    print(json.dumps(row), flush=True)
