#!/usr/bin/env python
"""grid_pull at config 2 through its organisations: routed (the default: a probe picks), sample tiles (pull_sorted), bricks of the
image (own_gather, INTERPOL_FLAG_BINNED_SCATTER); i.i.d. noise of several sigmas and a folding smooth field.
usage: tools/time_pull_router.py [check]   (check: parity of the bricks against the generic kernels on a spread of problems first)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
import bench
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, batch=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort(); return ts[len(ts) // 2]
TILES = _hip.FLAG_FORCE_TILED
BRICKS = _hip.FLAG_BINNED_SCATTER
if len(sys.argv) > 1 and sys.argv[1] == "check":
    g = torch.Generator().manual_seed(11)
    worst, nfail, n = 0.0, 0, 0
    for (ishape, oshape) in [((48, 48, 48), (48, 48, 48)), ((40, 33, 50), (37, 45, 29)), ((64, 64, 64), (40, 40, 40))]:
        for sigma in (0.0, 2.0, 7.0):
            for bound in range(7):
                for order in (3, 2):
                    ex = (bound + order) % 3
                    C = 1 + (bound + order) % 3
                    inp = torch.randn([2, C, *ishape], generator=g).to(dev)
                    lin = [torch.linspace(-2, n_ + 1, m) for n_, m in zip(ishape, oshape)]
                    grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + sigma * torch.randn([2, *oshape, 3], generator=g)).to(dev)
                    b = [bound, (bound + 3) % 7, (bound + 5) % 7]
                    ref = _hip.gather("pull", inp, grid, b, [order] * 3, ex, flags=_hip.FLAG_NO_FASTPATH)
                    for name, fl in (("bricks", BRICKS), ("routed", 0)):
                        got = _hip.gather("pull", inp, grid, b, [order] * 3, ex, flags=fl)
                        e = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
                        worst = max(worst, e); n += 1
                        if not e < 5e-6:
                            nfail += 1; print("FAIL", name, ishape, oshape, sigma, b, order, ex, C, e)
    print("parity cases", n, "failures", nfail, "worst rel err vs generic", worst)
res = {}
for name in ["sigma2", "sigma0", "sigma3", "sigma4", "sigma6", "smooth_amp8"]:
    if name == "smooth_amp8":
        grid = bench.smooth_grid(4, 256, 8.0, dev, 7); inp = torch.randn([4, 2, 256, 256, 256], device=dev)
    else:
        inp, grid = bench.make_inputs(4, 2, 256, float(name[5:]), dev, 1234)
    f = lambda fl: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    a = f(0); bb = f(BRICKS)
    res[name] = {"routed": round(timeit(lambda: f(0)), 3), "tiles": round(timeit(lambda: f(TILES)), 3), "bricks": round(timeit(lambda: f(BRICKS)), 3),
                 "maxdiff_routed_bricks": float((a - bb).abs().max())}
    del inp, grid, a, bb
print(json.dumps(res, indent=1))
