#!/usr/bin/env python
"""The gathering adjoint passes (csrc/resample1d.hip: resample1d_adj_gather) against the D-dimensional push on the separable lattice:
parity over dims / orders / bounds / extrapolation / lattices (monotone, constant runs, unsorted), and config-2-like timings of restrict."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import ops, separable
from interpol.sepgrid import SeparableGrid
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
g = torch.Generator(device=dev).manual_seed(9)
bad = 0
def lattice(kind, ns, nl):
    if kind == "affine":
        return torch.linspace(-0.7, nl - 0.4, ns, device=dev)
    if kind == "wide":
        return torch.linspace(-3.3, nl + 2.6, ns, device=dev)
    if kind == "runs":
        return torch.linspace(0.2, nl - 1.1, ns, device=dev).round()
    if kind == "unsorted":
        return torch.linspace(-0.5, nl - 0.5, ns, device=dev)[torch.randperm(ns, generator=g, device=dev)]
for dt in (torch.float32, torch.float64):
    for D, sshape, tshape in ((3, (70, 64, 130), (33, 40, 61)), (2, (150, 260), (64, 100)), (1, (3000,), (1100,)), (3, (20, 64, 64), (40, 30, 70))):
        for kind in ("affine", "wide", "runs", "unsorted"):
            for order in (0, 1, 2, 3, 5):
                for bound in range(7):
                    ex = (order + bound) % 3
                    x = torch.randn(2, 3, *sshape, generator=g, device=dev, dtype=dt)
                    lin = [lattice(kind, ns, nl).to(dt) for ns, nl in zip(sshape, tshape)]
                    o, b = [order] * D, [bound] * D
                    got = separable._SepPush.apply(x, lin, list(tshape), o, b, ex)
                    ref = ops.grid_push(x, SeparableGrid(lin), list(tshape), b, o, ex)
                    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
                    if not err < (2e-5 if dt == torch.float32 else 1e-12):
                        bad += 1
                        print("BAD", dt, D, kind, order, bound, ex, err, flush=True)
print("parity: bad =", bad, flush=True)
y = torch.randn(4, 2, 256, 256, 256, generator=g, device=dev)
res = {}
for order in (1, 3):
    res["restrict_o%d_ms" % order] = round(timeit(lambda: interpol.restrict(y, factor=[2, 2, 2], anchor='e', interpolation=order, bound='dct2')), 3)
x = torch.randn(4, 2, 128, 128, 128, generator=g, device=dev, requires_grad=True)
def fb():
    z = interpol.resize(x, factor=[2, 2, 2], anchor='e', interpolation=3, bound='dct2', prefilter=False)
    z.backward(y)
res["resize_fwd_bwd_o3_ms"] = round(timeit(fb), 3)
print(json.dumps(res), flush=True)
