#!/usr/bin/env python
"""gather5 / gather7 with the class-sorted queue of a brick's records (round 6) against list order (debug bit 64): BASELINE config 3
(8 x 1 x 192^3 fp32, order 5, dft, identity + N(0, sigma^2)) pull and grid_grad;
orders 4 - 7 at config 2's shape (4 x 2 x 256^3, dct2).  ms per call through interpol/_hip.py.  argv: [sigma]."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, bench, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0

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

def rows(tag, inp, grid, bound, orders):
    for order in orders:
        o = [order] * 3
        row = {"case": tag, "order": order, "sigma": sigma}
        # (explicit routing flag: debug bits switch the Python layer's automatic routing off)
        for name, fl in (("queue", _hip.FLAG_AUTO_SCATTER), ("list", _hip.FLAG_AUTO_SCATTER | (64 << 8))):
            row["pull_" + name] = timeit(lambda: _hip.gather("pull", inp, grid, bound, o, 1, flags=fl))
            row["grad_" + name] = timeit(lambda: _hip.gather("grad", inp, grid, bound, o, 1, flags=fl))
        print(json.dumps(row), flush=True)

g = torch.Generator(device=dev).manual_seed(3)
x3 = torch.randn(8, 1, 192, 192, 192, generator=g, device=dev)
g3 = torch.randn([8, 192, 192, 192, 3], generator=g, device=dev).mul_(sigma) + interpol.identity_grid([192] * 3, device=dev)
rows("cfg3 8x1x192^3 dft", x3, g3, [6, 6, 6], (5, 4))
del x3, g3
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
rows("cfg2 shape 4x2x256^3 dct2", inp, grid, [3, 3, 3], (5, 7))
