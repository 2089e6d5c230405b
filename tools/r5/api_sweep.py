#!/usr/bin/env python
"""A sweep over the API at config-2-like sizes: ms and bytes/s of algorithmic traffic per call, to spot operators far from their floor."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
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
g = torch.Generator(device=dev).manual_seed(1)
def row(name, fn, nbytes):
    ms = timeit(fn)
    print("%-58s %8.3f ms  %6.2f TB/s" % (name, ms, nbytes / ms / 1e9), flush=True)
B, C, n = 4, 2, 256
x = torch.randn(B, C, n, n, n, generator=g, device=dev)
ident = interpol.identity_grid([n, n, n], device=dev)[None]
grid = (ident + 2.0 * torch.randn(B, n, n, n, 3, generator=g, device=dev)).contiguous()
vox = B * n ** 3
io = vox * (12 + 2 * C * 4)
for order in (0, 1, 2, 3, 4, 5, 7):
    for bound in ("dct2", "zero"):
        kw = dict(interpolation=order, bound=bound, extrapolate=True)
        row("pull o%d %s" % (order, bound), lambda: interpol.grid_pull(x, grid, **kw), io)
        if bound == "dct2":
            row("push o%d %s" % (order, bound), lambda: interpol.grid_push(x, grid, **kw), io)
            row("count o%d" % order, lambda: interpol.grid_count(grid, **kw), vox * 16)
            if order >= 1:
                row("grad o%d" % order, lambda: interpol.grid_grad(x, grid, **kw), vox * (12 + C * 4 + 3 * C * 4))
for order in (2, 3, 5):
    row("spline_coeff_nd 3-D o%d dct2" % order, lambda: interpol.spline_coeff_nd(x, order, "dct2", 3), 2 * 3 * B * C * n ** 3 * 4)
row("mixed orders [1,2,3] pull", lambda: interpol.grid_pull(x, grid, interpolation=[1, 2, 3], bound="dct2", extrapolate=True), io)
row("mixed orders [1,2,3] push", lambda: interpol.grid_push(x, grid, interpolation=[1, 2, 3], bound="dct2", extrapolate=True), io)
row("pull o3 extrapolate=False", lambda: interpol.grid_pull(x, grid, interpolation=3, bound="dct2", extrapolate=False), io)
row("pull o1 prefilter (no-op)", lambda: interpol.grid_pull(x, grid, interpolation=1, bound="dct2", extrapolate=True, prefilter=True), io)
row("pull o3 prefilter=True", lambda: interpol.grid_pull(x, grid, interpolation=3, bound="dct2", extrapolate=True, prefilter=True), io)
xh = x.half()
row("pull o3 f16", lambda: interpol.grid_pull(xh, grid, interpolation=3, bound="dct2", extrapolate=True), vox * (12 + 2 * C * 2))
row("push o3 f16", lambda: interpol.grid_push(xh, grid, interpolation=3, bound="dct2", extrapolate=True), vox * (12 + 2 * C * 2))
x1 = torch.randn(64, 4, 1 << 16, generator=g, device=dev)
g1 = (torch.arange(1 << 16, device=dev, dtype=torch.float32)[None, :, None] + torch.randn(64, 1 << 16, 1, generator=g, device=dev)).contiguous()
row("1-D pull o3 64x4x65536", lambda: interpol.grid_pull(x1, g1, interpolation=3, bound="dct2", extrapolate=True), 64 * 65536 * (4 + 32))
row("1-D push o3 64x4x65536", lambda: interpol.grid_push(x1, g1, interpolation=3, bound="dct2", extrapolate=True), 64 * 65536 * (4 + 32))
row("identity_grid 256^3", lambda: interpol.identity_grid([n, n, n], device=dev), n ** 3 * 12)
row("resize 2x o1 128->256", lambda: interpol.resize(x[..., :128, :128, :128].contiguous(), factor=[2, 2, 2], anchor='e', interpolation=1, bound='dct2'), B * C * (128 ** 3 + n ** 3) * 4)
row("restrict 2x o3 256->128", lambda: interpol.restrict(x, factor=[2, 2, 2], anchor='e', interpolation=3, bound='dct2'), B * C * (128 ** 3 + n ** 3) * 4)
