#!/usr/bin/env python
"""BASELINE config 4 on one GPU: 64 (and 8, 16, 32) sources of 1x128^3 into a shared 512^3 target, cubic, replicate, push + count:
time, and the result against the target-stationary splatting of round 1 (interpol_push_bricks; backend.rough_deformations = False)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import backend
from interpol.distributed import push_count_shared
dev = torch.device("cuda", 0)
def timeit(fn, reps=3, inner=2):
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
g = torch.Generator(device=dev).manual_seed(4)
n, m = 128, 512
for nsrc in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    x = torch.randn(nsrc, 1, n, n, n, generator=g, device=dev)
    gr = torch.randn([nsrc, n, n, n, 3], generator=g, device=dev).mul_(2.0)
    gr += interpol.identity_grid([n, n, n], device=dev) * ((m - 1) / (n - 1))
    f = lambda: push_count_shared(x, gr, [m, m, m], interpolation=3, bound="replicate", extrapolate=True, reduce="none")
    res = {"nsrc": nsrc}
    backend.rough_deformations = None
    a = f(); res["ms_default"] = round(timeit(f), 3)
    backend.rough_deformations = False
    b = f(); res["ms_push_bricks"] = round(timeit(f), 3)
    backend.rough_deformations = None
    for name, u, v in (("push", a[0], b[0]), ("count", a[1], b[1])):
        res[name + "_rel_diff"] = "%.1e" % float((u - v).abs().max() / v.abs().max())
    print(json.dumps(res), flush=True)
    del x, gr, a, b
