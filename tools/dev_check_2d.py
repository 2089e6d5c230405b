#!/usr/bin/env python
"""Development check of the lean 2-D tiles (ops_tiled2d.hip) against the generic kernels, + timing at config 5."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
torch.manual_seed(3)
res = {}
for (B, C, shp, gshp) in [(2, 3, (70, 90), (70, 90)), (1, 5, (40, 130), (100, 77)), (3, 1, (64, 64), (64, 64))]:
    x = torch.randn(B, C, *shp, device=dev)
    v = torch.randn(B, C, *gshp, device=dev)
    for sigma in (0.0, 2.0, 12.0):
        g = interpol.identity_grid(gshp, device=dev)[None] * (torch.tensor(shp, device=dev) - 1) / (torch.tensor(gshp, device=dev) - 1) \
            + sigma * torch.randn(B, *gshp, 2, device=dev)
        for o0 in (1, 2, 3):
            for o1 in (1, 3):
                for bound in range(7):
                    ex = (o0 + bound) % 3
                    b, o = [bound, (bound + 2) % 7], [o0, o1]
                    for dt, tol in ((torch.float32, 3e-6), (torch.bfloat16, 1e-2)):
                        a = _hip.gather("pull", x.to(dt), g, b, o, ex)
                        r = _hip.gather("pull", x.to(dt), g, b, o, ex, flags=_hip.FLAG_NO_FASTPATH)
                        e = relerr(a.float(), r.float()); res["pull_" + str(dt)] = max(res.get("pull_" + str(dt), 0), e)
                        if e > tol: print("MISMATCH pull", B, C, shp, gshp, sigma, o, b, ex, dt, e)
                        a = _hip.scatter("push", v.to(dt), g, list(shp), b, o, ex, with_count=True)
                        r = _hip.scatter("push", v.to(dt), g, list(shp), b, o, ex, with_count=True, flags=_hip.FLAG_NO_FASTPATH)
                        e = relerr(a.float(), r.float()); res["push_" + str(dt)] = max(res.get("push_" + str(dt), 0), e)
                        if e > max(tol, 1e-5): print("MISMATCH push", B, C, shp, gshp, sigma, o, b, ex, dt, e)
                    a = _hip.scatter("count", None, g, list(shp), b, o, ex)
                    r = _hip.scatter("count", None, g, list(shp), b, o, ex, flags=_hip.FLAG_NO_FASTPATH)
                    e = relerr(a, r); res["count"] = max(res.get("count", 0), e)
                    if e > 1e-5: print("MISMATCH count", sigma, o, b, ex, e)
print(json.dumps(res))
B, C, n = 32, 3, 1024
gen = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, C, n, n, generator=gen, device=dev).to(torch.bfloat16)
for sigma in (2.0, 0.0):
    gr = torch.randn([B, n, n, 2], generator=gen, device=dev).mul_(sigma) + interpol.identity_grid([n, n], device=dev)
    kw = dict(interpolation=[2, 3], bound=["dct1", "dst2"], extrapolate=True)
    t = {"pull_bf16": timeit(lambda: interpol.grid_pull(x, gr, **kw)), "push_bf16": timeit(lambda: interpol.grid_push(x, gr, **kw)),
         "pull_f32": timeit(lambda: interpol.grid_pull(x.float(), gr, **kw)), "push_f32": timeit(lambda: interpol.grid_push(x.float(), gr, **kw)),
         "pull_bf16_generic": timeit(lambda: _hip.gather("pull", x, gr, [2, 5], [2, 3], 1, flags=1)),
         "push_bf16_r1": timeit(lambda: _hip.scatter("push", x, gr, None, [2, 5], [2, 3], 1, flags=32 << 8))}
    print("sigma", sigma, json.dumps({k: round(v, 3) for k, v in t.items()}))
