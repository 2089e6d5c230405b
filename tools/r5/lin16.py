import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
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
for dt, tol in ((torch.bfloat16, 8e-3), (torch.float16, 1e-3)):
    for shape, oshape in (((40, 33, 50), (37, 45, 29)), ((64, 64, 64), (64, 64, 64))):
        for bound in range(7):
            for sigma in (0.05, 5.0):
                ex = bound % 3
                img = torch.randn(2, 3, *shape, generator=g, device=dev).to(dt)
                lin = [torch.linspace(-2, n + 1, m, device=dev) for n, m in zip(shape, oshape)]
                grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + sigma * torch.randn(2, *oshape, 3, generator=g, device=dev)).contiguous()
                b = [bound, (bound + 3) % 7, (bound + 5) % 7]
                r = _hip.gather("pull", img.float(), grid, b, [1] * 3, ex, flags=_hip.FLAG_NO_FASTPATH)
                for fl in (0, _hip.FLAG_BINNED_SCATTER):
                    a = _hip.gather("pull", img, grid, b, [1] * 3, ex, flags=fl)
                    e = float((a.float() - r).abs().max() / r.abs().max())
                    if not (e < tol and a.dtype == dt):
                        bad += 1; print("BAD", dt, shape, bound, sigma, fl, e, flush=True)
print("parity: bad =", bad, flush=True)
B, C, n = 4, 2, 256
ident = interpol.identity_grid([n, n, n], device=dev)[None]
x = torch.randn(B, C, n, n, n, generator=g, device=dev).bfloat16()
for s in (0.0, 2.0):
    grid = (ident + s * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
    print(json.dumps({"sigma": s, "bf16_generic": round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1, flags=_hip.FLAG_NO_FASTPATH)), 3),
                      "bf16_default": round(timeit(lambda: _hip.gather("pull", x, grid, [3] * 3, [1] * 3, 1)), 3)}), flush=True)
