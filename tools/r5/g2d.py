#!/usr/bin/env python
"""2-D grid_pull / backward through bricks of the image (scatter2d.hip: gather2d) against the lean tiles and the generic kernels."""
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
bad = 0
def rel(a, r):
    return float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-30))
if "parity" in sys.argv or len(sys.argv) == 1:
    for (B, C, ny, nz, oy, oz) in [(2, 3, 100, 131, 97, 140), (1, 1, 64, 64, 64, 64), (2, 5, 150, 129, 160, 140)]:
        ident = interpol.identity_grid([oy, oz], device=dev)[None]
        scale = torch.tensor([ny / oy, nz / oz], device=dev)
        for bound in [[0, 0], [1, 2], [3, 4], [5, 6], [2, 5], [6, 0]]:
            for order in [[1, 1], [2, 3], [3, 3], [3, 1], [2, 2], [1, 2]]:
                for dt in [torch.float32, torch.bfloat16, torch.float16]:
                    for sigma in [1.0, 20.0]:
                        for extra in [1, 0]:
                            img = torch.randn(B, C, ny, nz, generator=g, device=dev).to(dt)
                            gout = torch.randn(B, C, oy, oz, generator=g, device=dev).to(dt)
                            grid = (ident * scale + sigma * torch.randn(B, oy, oz, 2, generator=g, device=dev)).contiguous()
                            tol = 4e-6 if dt == torch.float32 else (8e-3 if dt == torch.bfloat16 else 1e-3)
                            errs = []
                            for rd in (True, None):
                                backend.rough_deformations = rd
                                a = _hip.gather("pull", img, grid, bound, order, extra)
                                gv, gg = _hip.pull_backward(gout, img, grid, bound, order, extra, True, True)
                                _, gg1 = _hip.pull_backward(gout, img, grid, bound, order, extra, False, True)
                                pv, pg = _hip.push_backward(img, gout, grid, bound, order, extra, True, True)
                                _, cg = _hip.push_backward(img[:, :1].contiguous(), None, grid, bound, order, extra, False, True)
                                r = _hip.gather("pull", img.float(), grid, bound, order, extra, flags=_hip.FLAG_NO_FASTPATH)
                                rv, rg = _hip.pull_backward(gout.float(), img.float(), grid, bound, order, extra, True, True, flags=_hip.FLAG_NO_FASTPATH)
                                qv, qg = _hip.push_backward(img.float(), gout.float(), grid, bound, order, extra, True, True, flags=_hip.FLAG_NO_FASTPATH)
                                _, dg = _hip.push_backward(img[:, :1].contiguous().float(), None, grid, bound, order, extra, False, True, flags=_hip.FLAG_NO_FASTPATH)
                                errs += [rel(a, r), rel(gv, rv), rel(gg, rg) / 4, rel(gg1, rg) / 4, rel(pv, qv), rel(pg, qg) / 4, rel(cg, dg) / 4]
                            backend.rough_deformations = None
                            if not all(e < tol for e in errs):
                                bad += 1
                                print("BAD", B, C, ny, nz, bound, order, dt, sigma, extra, ["%.1e" % e for e in errs], flush=True)
    print("parity: bad =", bad, flush=True)
if "time" in sys.argv or len(sys.argv) == 1:
    B, C, n = 32, 3, 1024
    x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
    ident = interpol.identity_grid([n, n], device=dev)[None]
    bc, o = [2, 5], [2, 3]
    for sigma in [float(a) for a in os.environ.get("S2D_SIGMAS", "0,2,4,8,16").split(",")]:
        grid = (ident + sigma * torch.randn(B, n, n, 2, generator=g, device=dev)).contiguous()
        res = {"sigma": sigma}
        for name, rd in (("bricks", True), ("tiles", False), ("auto", None)):
            backend.rough_deformations = rd
            res["pull_" + name] = round(timeit(lambda: _hip.gather("pull", x, grid, bc, o, 1)), 3)
            res["ggrid_" + name] = round(timeit(lambda: _hip.pull_backward(x, x, grid, bc, o, 1, False, True)), 3)
            res["bwd_" + name] = round(timeit(lambda: _hip.pull_backward(x, x, grid, bc, o, 1, True, True)), 3)
        backend.rough_deformations = None
        print(json.dumps(res), flush=True)
        del grid
