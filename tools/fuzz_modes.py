#!/usr/bin/env python
"""Differential fuzz of the coordinate modes and scatter variants through the default routing vs the generic kernels:
displacement fields (INTERPOL_FLAG_DISPLACEMENT), affine lattices (interpol.AffineGrid), batch-broadcast grids,
push with the count channel, shared targets.  usage: tools/fuzz_modes.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
GEN = _hip.FLAG_NO_FASTPATH
def rel(a, r):
    a, r = a.float(), r.float()
    return float((a - r).abs().max() / r.abs().max().clamp_min(1e-20))
bad = 0
for case in range(n_cases):
    dim = rnd.choice([2, 3, 3])
    B, C = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3])
    shp = [rnd.randint(20, 70) for _ in range(3)] if dim == 3 else [rnd.randint(40, 300) for _ in range(2)]
    order = [rnd.choice([1, 2, 3])] * dim
    bound = [rnd.randrange(7)] * dim
    ex = rnd.choice([1, 1, 0, 2])
    mode = rnd.choice(["displacement", "affine", "broadcast", "with_count", "shared"])
    vol = torch.randn([B, C, *shp], generator=gen).to(dev)
    src = torch.randn([B, C, *shp], generator=gen).to(dev)
    zoom = rnd.choice([1.0, 1.0, 1.6, 2.3])
    checks = []
    try:
        if mode == "displacement":
            ident = interpol.identity_grid(shp)
            disp = ((zoom - 1.0) * (ident - ident.reshape(-1, dim).mean(0)) + rnd.choice([0.0, 1.0]) * torch.randn([B, *shp, dim], generator=gen)).contiguous().to(dev)
            fl = _hip.FLAG_DISPLACEMENT
            checks.append(("pull", rel(_hip.gather("pull", vol, disp, bound, order, ex, flags=fl), _hip.gather("pull", vol, disp, bound, order, ex, flags=fl | GEN))))
            checks.append(("grad", rel(_hip.gather("grad", vol, disp, bound, order, ex, flags=fl), _hip.gather("grad", vol, disp, bound, order, ex, flags=fl | GEN))))
            checks.append(("push", rel(_hip.scatter("push", src, disp, shp, bound, order, ex, flags=fl), _hip.scatter("push", src, disp, shp, bound, order, ex, flags=fl | GEN))))
            a = _hip.pull_backward(src, vol, disp, bound, order, ex, True, True, flags=fl); r = _hip.pull_backward(src, vol, disp, bound, order, ex, True, True, flags=fl | GEN)
            checks += [("pullbwd_%d" % i, rel(x, y)) for i, (x, y) in enumerate(zip(a, r))]
            a = _hip.push_backward(vol, src, disp, bound, order, ex, True, True, flags=fl); r = _hip.push_backward(vol, src, disp, bound, order, ex, True, True, flags=fl | GEN)
            checks += [("pushbwd_%d" % i, rel(x, y)) for i, (x, y) in enumerate(zip(a, r))]
        elif mode == "affine":
            mat = torch.eye(dim, dim + 1) * zoom + 0.05 * torch.randn([dim, dim + 1], generator=gen)
            mat[:, -1] = torch.tensor([rnd.uniform(-3, 3) for _ in range(dim)])
            ag = interpol.AffineGrid(mat.to(dev), shp)
            checks.append(("pull", rel(_hip.gather("pull", vol, ag, bound, order, ex), _hip.gather("pull", vol, ag, bound, order, ex, flags=GEN))))
            checks.append(("grad", rel(_hip.gather("grad", vol, ag, bound, order, ex), _hip.gather("grad", vol, ag, bound, order, ex, flags=GEN))))
            checks.append(("push", rel(_hip.scatter("push", src, ag, shp, bound, order, ex), _hip.scatter("push", src, ag, shp, bound, order, ex, flags=GEN))))
        else:
            ident = interpol.identity_grid(shp)
            gb = 1 if mode == "broadcast" else B
            grid = ((ident - ident.reshape(-1, dim).mean(0)) * zoom + ident.reshape(-1, dim).mean(0) + 0.5 * torch.randn([gb, *shp, dim], generator=gen)).contiguous().to(dev)
            if mode == "broadcast":
                checks.append(("pull", rel(_hip.gather("pull", vol, grid, bound, order, ex), _hip.gather("pull", vol, grid, bound, order, ex, flags=GEN))))
                checks.append(("grad", rel(_hip.gather("grad", vol, grid, bound, order, ex), _hip.gather("grad", vol, grid, bound, order, ex, flags=GEN))))
                checks.append(("push", rel(_hip.scatter("push", src, grid, shp, bound, order, ex), _hip.scatter("push", src, grid, shp, bound, order, ex, flags=GEN))))
            elif mode == "with_count":
                checks.append(("push+count", rel(_hip.scatter("push", src, grid, shp, bound, order, ex, with_count=True), _hip.scatter("push", src, grid, shp, bound, order, ex, flags=GEN, with_count=True))))
            else:
                checks.append(("shared", rel(_hip.scatter("push", src, grid, shp, bound, order, ex, shared=True), _hip.scatter("push", src, grid, shp, bound, order, ex, flags=GEN, shared=True))))
                checks.append(("shared+count", rel(_hip.scatter("push", src, grid, shp, bound, order, ex, shared=True, with_count=True), _hip.scatter("push", src, grid, shp, bound, order, ex, flags=GEN, shared=True, with_count=True))))
    except Exception as e:
        print("EXCEPTION", case, mode, dim, B, C, shp, order, bound, ex, zoom, repr(e)); bad += 1; continue
    fails = [(k, "%.1e" % v) for k, v in checks if not v <= (5e-4 if zoom > 1.5 else 3e-5)]
    if fails:
        bad += 1
        print("MISMATCH case", case, mode, "dim", dim, "B", B, "C", C, shp, "order", order, "bound", bound, "ex", ex, "zoom", zoom, fails)
print("fuzz modes: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
