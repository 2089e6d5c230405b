#!/usr/bin/env python
"""What probe2d sees at config 5's shape: pixels per thousand outside the lean tiles' boxes, by sigma, next to the tiles' and the bricks' times."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=3):
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
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 32, 3, 1024
x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
ident = interpol.identity_grid([n, n], device=dev)[None]
bc, o = [2, 5], [2, 3]
fields = [("sigma %g" % s, lambda s=s: ident + s * torch.randn(B, n, n, 2, generator=g, device=dev)) for s in (2, 3, 4, 4.5, 5, 5.5, 6, 8)]
fields += [("zoom %g" % z, lambda z=z: (ident - n / 2) * z + n / 2) for z in (1.5, 2.0, 2.2, 2.5, 3.0)]
yy = torch.arange(n, device=dev, dtype=torch.float32)
fields += [("smooth amplitude %g" % a, lambda a=a: ident + a * torch.stack([torch.sin(yy[:, None] / 17) * torch.cos(yy[None, :] / 23), torch.cos(yy[:, None] / 19) * torch.sin(yy[None, :] / 13)], -1)[None]) for a in (8, 24, 48)]
for name, mk in fields:
    grid = mk().expand(B, n, n, 2).contiguous()
    backend.rough_deformations = None
    _hip.gather("pull", x, grid, bc, o, 1)
    torch.cuda.synchronize()
    ws = list(_hip._WS_CACHE.values())[0]
    ws = ws[0] if isinstance(ws, (tuple, list)) else ws
    word = int(ws.view(torch.uint8)[136:144].view(torch.int64).item()) & 0xffffffffffffffff
    verdict = int(ws.view(torch.uint8)[128:132].view(torch.int32).item())
    out, es, nt = word >> 41, (word >> 22) & 0x7ffff, (word >> 10) & 0xfff
    res = {"field": name, "ppm_outside": round(1e6 * out / max(nt * 1024, 1), 1), "density": round(es / 32.0 / max(nt, 1), 3), "gather_verdict": verdict}
    for nm, rd in (("tiles", False), ("bricks", True), ("auto", None)):
        backend.rough_deformations = rd
        res["pull_" + nm] = round(timeit(lambda: _hip.gather("pull", x, grid, bc, o, 1)), 3)
        res["push_" + nm] = round(timeit(lambda: _hip.scatter("push", x, grid, None, bc, o, 1)), 3)
    backend.rough_deformations = None
    print(json.dumps(res), flush=True)
    del grid
