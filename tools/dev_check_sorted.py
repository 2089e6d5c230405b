#!/usr/bin/env python
"""Development check on the GPU box: class-sorted tiles vs the generic kernels (and timing vs the
natural-order tiles).  usage: dev_check_sorted.py [op ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench

dev = torch.device("cuda", 0)

def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())

res = {}
# correctness at odd sizes, all bounds, both orders, extrapolate modes
torch.manual_seed(7)
for (B, C, shp, gshp) in [(2, 2, (33, 38, 45), (33, 38, 45)), (1, 3, (20, 50, 17), (40, 23, 31)), (2, 1, (64, 64, 64), (64, 64, 64))]:
    x = torch.randn(B, C, *shp, device=dev)
    for sigma in (0.0, 0.7, 2.0, 6.0):
        g = interpol.identity_grid(gshp, device=dev)[None] * (torch.tensor(shp, device=dev) - 1) / (torch.tensor(gshp, device=dev) - 1) \
            + sigma * torch.randn(B, *gshp, 3, device=dev)
        for order in (2, 3):
            for bound in range(7):
                for ext in (0, 1, 2):
                    a = _hip.gather("pull", x, g, [bound] * 3, [order] * 3, ext)
                    r = _hip.gather("pull", x, g, [bound] * 3, [order] * 3, ext, flags=_hip.FLAG_NO_FASTPATH)
                    e = relerr(a, r)
                    key = "pull_max_relerr_vs_generic"
                    res[key] = max(res.get(key, 0.0), e)
                    if e > 2e-6:
                        print("MISMATCH", B, C, shp, gshp, sigma, order, bound, ext, e)
                    if "push" in sys.argv:
                        v = torch.randn(B, C, *gshp, device=dev)
                        for op, args in (("push", (v,)), ("count", ())):
                            a = _hip.scatter(op, v if op == "push" else None, g, list(shp), [bound] * 3, [order] * 3, ext)
                            r = _hip.scatter(op, v if op == "push" else None, g, list(shp), [bound] * 3, [order] * 3, ext, flags=_hip.FLAG_NO_FASTPATH)
                            e = relerr(a, r)
                            key = op + "_max_relerr_vs_generic"
                            res[key] = max(res.get(key, 0.0), e)
                            if e > 1e-5:
                                print("MISMATCH", op, B, C, shp, gshp, sigma, order, bound, ext, e)
                        a = _hip.scatter("push", v, g, list(shp), [bound] * 3, [order] * 3, ext, with_count=True)
                        r = _hip.scatter("push", v, g, list(shp), [bound] * 3, [order] * 3, ext, with_count=True, flags=_hip.FLAG_NO_FASTPATH)
                        a = a if torch.is_tensor(a) else torch.cat([x.flatten() for x in a]); r = r if torch.is_tensor(r) else torch.cat([x.flatten() for x in r])
                        e = relerr(a, r)
                        res["pushcount_max_relerr_vs_generic"] = max(res.get("pushcount_max_relerr_vs_generic", 0.0), e)
                        if e > 1e-5:
                            print("MISMATCH pushcount", B, C, shp, gshp, sigma, order, bound, ext, e)
print(json.dumps(res))
# timing at config 2
B, C, n = 4, 2, 256
for sigma in (2.0, 0.0):
    inp, grid = bench.make_inputs(B, C, n, sigma, dev, 1234)
    t = {}
    t["pull_sorted"] = timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1))
    t["pull_tiled_r1"] = timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=32 << 8))
    a = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)
    r = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=32 << 8)
    t["relerr_vs_r1"] = relerr(a, r)
    if "push" in sys.argv:
        t["push_binned"] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1))
        t["push_sorted"] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=(32 | 128) << 8))
        t["push_tiled_r1"] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=32 << 8))
        a = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1)
        r = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=32 << 8)
        t["count_binned"] = timeit(lambda: _hip.scatter("count", None, grid, [n] * 3, [3] * 3, [3] * 3, 1))
        t["push_relerr_vs_r1"] = relerr(a, r)
        t["count_sorted"] = timeit(lambda: _hip.scatter("count", None, grid, [n] * 3, [3] * 3, [3] * 3, 1))
        t["count_tiled_r1"] = timeit(lambda: _hip.scatter("count", None, grid, [n] * 3, [3] * 3, [3] * 3, 1, flags=32 << 8))
    print("sigma", sigma, json.dumps({k: float("%.4g" % v) for k, v in t.items()}))
