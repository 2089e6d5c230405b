#!/usr/bin/env python
"""Trilinear grid_pull: the generic kernel (the default of rounds 1-4) against the class-sorted LDS tiles with K = 1 (FLAG_FORCE_TILED)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
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
g = torch.Generator(device=dev).manual_seed(2)
bad = 0
for shape, oshape in (((40, 33, 50), (37, 45, 29)), ((64, 64, 64), (64, 64, 64)), ((30, 40, 36), (48, 40, 52))):
    for bound in range(7):
        for ex in (0, 1, 2):
            for sigma in (0.5, 5.0):
                img = torch.randn(2, 3, *shape, generator=g, device=dev)
                lin = [torch.linspace(-2, n + 1, m, device=dev) for n, m in zip(shape, oshape)]
                grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + sigma * torch.randn(2, *oshape, 3, generator=g, device=dev)).contiguous()
                b = [bound, (bound + 3) % 7, (bound + 5) % 7]
                r = _hip.gather("pull", img, grid, b, [1] * 3, ex, flags=_hip.FLAG_NO_FASTPATH)
                e = 0.0
                for fl in (_hip.FLAG_FORCE_TILED, 0, _hip.FLAG_BINNED_SCATTER):
                    a = _hip.gather("pull", img, grid, b, [1] * 3, ex, flags=fl)
                    e = max(e, float((a - r).abs().max() / r.abs().max()))
                gout = torch.randn(2, 3, *oshape, generator=g, device=dev)
                rg = _hip.pull_backward(gout, img, grid, b, [1] * 3, ex, False, True, flags=_hip.FLAG_NO_FASTPATH)[1]
                rp = _hip.push_backward(img, gout, grid, b, [1] * 3, ex, True, True, flags=_hip.FLAG_NO_FASTPATH)
                for fl in (_hip.FLAG_FORCE_TILED, 0, _hip.FLAG_BINNED_SCATTER):
                    ag = _hip.pull_backward(gout, img, grid, b, [1] * 3, ex, False, True, flags=fl)[1]
                    e = max(e, float((ag - rg).abs().max() / rg.abs().max()) / 4)
                    if fl != _hip.FLAG_FORCE_TILED:
                        ab = _hip.pull_backward(gout, img, grid, b, [1] * 3, ex, True, True, flags=fl)
                        e = max(e, float((ab[1] - rg).abs().max() / rg.abs().max()) / 4)
                        ap = _hip.push_backward(img, gout, grid, b, [1] * 3, ex, True, True, flags=fl)
                        e = max(e, float((ap[0] - rp[0]).abs().max() / rp[0].abs().max()), float((ap[1] - rp[1]).abs().max() / rp[1].abs().max()) / 4)
                if not e < 2e-6:
                    bad += 1; print("BAD", shape, bound, ex, sigma, e, flush=True)
print("parity: bad =", bad, flush=True)
B, C, n = 4, 2, 256
ident = interpol.identity_grid([n, n, n], device=dev)[None]
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
yy = torch.arange(n, device=dev, dtype=torch.float32)
smooth = torch.stack([torch.sin(yy[:, None, None] / 17) * torch.cos(yy[None, :, None] / 23) * torch.ones(n, device=dev)[None, None, :]] * 3, -1)[None]
for name, mk in [("sigma %g" % s, (lambda s=s: ident + s * torch.randn(B, n, n, n, 3, generator=g, device=dev))) for s in (0.0, 0.25, 0.5, 1.0, 2.0, 4.0)] + [("smooth amp 4", lambda: (ident + 4 * smooth).expand(B, n, n, n, 3))]:
    grid = mk().contiguous()
    res = {"field": name}
    res["generic_ms"] = round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1)), 3)
    res["generic_ms"] = round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1, flags=_hip.FLAG_NO_FASTPATH)), 3)
    res["sorted_k1_ms"] = round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)), 3)
    res["default_ms"] = round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1)), 3)
    print(json.dumps(res), flush=True)
    del grid
