import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
O = int(os.environ.get("LP_ORDER", "1"))      # 1: trilinear, 0: nearest neighbour (the bricks, always)
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
g = torch.Generator(device=dev).manual_seed(2)
bad = 0
for shape, oshape in (((40, 33, 50), (37, 45, 29)), ((64, 64, 64), (64, 64, 64)), ((48, 48, 48), (60, 50, 70))):
    for bound in range(7):
        for sigma in (0.5, 6.0):
            for dt in (torch.float32, torch.bfloat16):
                ex = bound % 3
                C = 1 + bound % 3
                src = torch.randn(2, C, *oshape, generator=g, device=dev).to(dt)
                lin = [torch.linspace(-2, n + 1, m, device=dev) for n, m in zip(shape, oshape)]
                grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + sigma * torch.randn(2, *oshape, 3, generator=g, device=dev)).contiguous()
                b = [bound, (bound + 3) % 7, (bound + 5) % 7]
                r = _hip.scatter("push", src.float(), grid, list(shape), b, [O] * 3, ex, flags=_hip.FLAG_NO_FASTPATH, with_count=True)
                rc = _hip.scatter("count", None, grid, list(shape), b, [O] * 3, ex, flags=_hip.FLAG_NO_FASTPATH)
                tol = 4e-6 if dt == torch.float32 else 8e-3
                for fl in (0, _hip.FLAG_BINNED_SCATTER):
                    a = _hip.scatter("push", src, grid, list(shape), b, [O] * 3, ex, flags=fl, with_count=True)
                    c = _hip.scatter("count", None, grid, list(shape), b, [O] * 3, ex, flags=fl)
                    e = max(float((a.float() - r).abs().max() / r.abs().max()), float((c - rc).abs().max() / rc.abs().max()) * (tol / 4e-6 if False else 1))
                    if not e < tol:
                        bad += 1; print("BAD", shape, bound, sigma, dt, fl, e, flush=True)
print("parity: bad =", bad, flush=True)
B, C, n = 4, 2, 256
ident = interpol.identity_grid([n, n, n], device=dev)[None]
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
for s in ((0.0, 0.1, 0.2, 0.3, 0.5, 1.0, 2.0, 6.0) if O == 0 else (0.0, 1.0, 2.0, 4.0, 6.0)):
    grid = (ident + s * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
    res = {"sigma": s}
    for name, rd in (("tiles", False), ("bricks", True), ("default", None)):
        backend.rough_deformations = rd
        res["push_" + name] = round(timeit(lambda: _hip.scatter("push", x, grid, None, [3] * 3, [O] * 3, 1)), 3)
    backend.rough_deformations = None
    res["gvol_default"] = round(timeit(lambda: _hip.pull_backward(x, x, grid, [3] * 3, [O] * 3, 1, True, False)), 3)
    print(json.dumps(res), flush=True)
    del grid
