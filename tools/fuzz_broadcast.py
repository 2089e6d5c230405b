#!/usr/bin/env python
"""Autograd-level fuzz of batch broadcasting (input batch 1 or B, grid batch 1 or B; nd.py:95): outputs and gradients of
grid_pull / grid_push / grid_grad must equal those of the explicitly expanded tensors (gradients of the broadcast operand
summed over the batch).  usage: tools/fuzz_broadcast.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); gen = torch.Generator().manual_seed(seed)
bad = 0
for case in range(n_cases):
    dim = rnd.choice([2, 3, 3]); B = rnd.choice([2, 3]); C = rnd.choice([1, 2, 3])
    shp = [rnd.randint(18, 48) for _ in range(3)] if dim == 3 else [rnd.randint(40, 160) for _ in range(2)]
    order = rnd.choice([1, 2, 3, 3, 5]); bound = rnd.choice(["dct2", "zero", "replicate", "dft", "dst2"]); ex = rnd.choice([True, True, False])
    zoom = rnd.choice([1.0, 1.0, 2.4]); bx, bg = rnd.choice([(1, B), (B, 1), (1, 1)])
    dt = rnd.choice([torch.float32, torch.float32, torch.bfloat16])
    op = rnd.choice(["pull", "push", "grad"])
    x0 = torch.randn([bx, C, *shp], generator=gen).to(dev).to(dt)
    g0 = (interpol.identity_grid(shp) * zoom + 0.2 * torch.randn([bg, *shp, dim], generator=gen)).to(dev)
    kw = dict(interpolation=order, bound=bound, extrapolate=ex)
    fn = {"pull": interpol.grid_pull, "push": interpol.grid_push, "grad": interpol.grid_grad}[op]
    res = []
    try:
        for expand in (False, True):
            x = (x0.expand(B, *x0.shape[1:]).contiguous() if expand else x0.clone()).requires_grad_(True)
            g = (g0.expand(B, *g0.shape[1:]).contiguous() if expand else g0.clone()).requires_grad_(True)
            if bx == 1 and bg == 1 and expand:
                pass
            y = fn(x, g, **kw)
            w = torch.randn(y.shape, generator=torch.Generator().manual_seed(case)).to(dev).to(y.dtype) if not res else res[0][3]
            if expand and y.shape != w.shape: w = w.expand(y.shape)
            (y.float() * w.float()).sum().backward()
            gx = x.grad.float(); gg = g.grad
            if expand and bx == 1: gx = gx.sum(0, keepdim=True)
            if expand and bg == 1: gg = gg.sum(0, keepdim=True)
            res.append((y.detach().float(), gx, gg, w))
    except Exception as e:
        print("EXCEPTION", case, op, dim, B, C, shp, order, bound, ex, zoom, (bx, bg), dt, repr(e)); bad += 1; continue
    ya, yb = res[0][0], res[1][0]
    if ya.shape != yb.shape: yb = yb[:ya.shape[0]]
    tol = 3e-2 if dt != torch.float32 else (1e-4 if order < 5 else 1e-3)
    errs = {"out": (ya, yb), "grad_input": (res[0][1], res[1][1]), "grad_grid": (res[0][2], res[1][2])}
    fails = []
    for k, (a, b) in errs.items():
        if bx == 1 and bg == 1 and k != "out":
            b = b / B if k == "grad_input" or k == "grad_grid" else b          # B identical items: the expanded loss counts each B times
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
        if not e <= tol: fails.append((k, "%.1e" % e))
    if fails:
        bad += 1; print("MISMATCH", case, op, "dim", dim, "B", B, "C", C, shp, "order", order, bound, ex, "zoom", zoom, "batches", (bx, bg), dt, fails)
print("fuzz broadcast: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
