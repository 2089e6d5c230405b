import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=4):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / inner)
    ts.sort(); return ts[len(ts) // 2]
g = torch.Generator().manual_seed(3)
n = 192
inp = torch.randn([8, 1, n, n, n], generator=g).to(dev)
ident = interpol.identity_grid([n] * 3)[None].expand(8, n, n, n, 3)
B, T = _hip.FLAG_BINNED_SCATTER, _hip.FLAG_FORCE_TILED
for name, grid in (("zoom1.2", (ident - 95.5) * 1.2 + 95.5), ("zoom1.5", (ident - 95.5) * 1.5 + 95.5), ("zoom2", (ident - 95.5) * 2 + 95.5), ("stride2_inside", ident * 0.5 + 20), ("zoom0.7", (ident - 95.5) * 0.7 + 95.5)):
    grid = grid.contiguous().to(dev)
    for order in (5, 4):
        res = {}
        for op in ("pull", "grad"):
            f = lambda fl=0: _hip.gather(op, inp, grid, [6] * 3, [order] * 3, 1, flags=fl)
            res[op] = (round(timeit(lambda: f(T)), 3), round(timeit(lambda: f(B)), 3), round(timeit(f), 3))
        print(name, "order", order, "(tiles, bricks, default)", res, flush=True)
