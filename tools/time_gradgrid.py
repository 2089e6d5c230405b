#!/usr/bin/env python
"""Grid gradient of the pull (interpol_pull_backward, grad_grid alone) at BASELINE config 2 (4 x 2 x 256^3 cubic dct2): the routed
default (sample tiles + bricks of the image for the tiles they flag), the sample tiles alone, the bricks alone -- on i.i.d. noise
of sigma voxels, the identity and a smooth field.  ms per call (median of 5 x 4 back-to-back calls)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)


def timeit(fn, reps=5, inner=4):
    for _ in range(3):
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


order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cases = {}
for s in (0.0, 0.5, 1.0, 2.0, 3.0, 4.0):
    cases["sigma_%g" % s] = bench.make_inputs(4, 2, 256, s, dev, 1234)[1]
cases["smooth"] = bench.smooth_grid(4, 256, 2.0, dev, 7)
inp = torch.randn([4, 2, 256, 256, 256], device=dev)
gout = torch.randn_like(inp)
for name, grid in cases.items():
    gf = lambda fl=0: _hip.pull_backward(gout, inp, grid, [3] * 3, [order] * 3, 1, False, True, flags=fl)[1]
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [order] * 3, 1, flags=fl)
    res = {}
    backend.rough_deformations = None
    res["gradgrid_routed"] = round(timeit(gf), 3)
    res["gradgrid_bricks"] = round(timeit(lambda: gf(_hip.FLAG_BINNED_SCATTER)), 3)
    res["pull_routed"] = round(timeit(pf), 3)
    res["pull_bricks"] = round(timeit(lambda: pf(_hip.FLAG_BINNED_SCATTER)), 3)
    backend.rough_deformations = False
    res["gradgrid_tiles"] = round(timeit(gf), 3)
    res["pull_tiles"] = round(timeit(pf), 3)
    backend.rough_deformations = None
    print(name, json.dumps(res), flush=True)
