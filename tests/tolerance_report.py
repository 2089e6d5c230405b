#!/usr/bin/env python
"""Worst error of the HIP fp32 path against the reference's fp64 golden outputs, in units of north_star's
tolerance (|err| <= 1e-5 |ref| + 1e-5 max|ref|), per operator and spline order; backward and prefilter vectors as
max|err| / (1e-5 max|ref|).  Run on the GPU box: python tests/tolerance_report.py  (writes a table to stdout)."""
import os, sys, json, collections
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import numpy as np, torch, interpol
import golden_util as G
from test_hip_parity import HipOps, DEV


def ratio(got, want, rtol=1e-5, atol_rel=1e-5):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    if want.size == 0:
        return 0.0
    atol = atol_rel * max(float(np.abs(want).max()), 1e-30)
    return float((np.abs(got - want) / (rtol * np.abs(want) + atol)).max())


worst = collections.defaultdict(float)
hip = HipOps(torch.float32)
for c in G.manifest()["cases"]:
    o = max(c["order"][:c["dim"]])
    got = G.run_case(hip, c, np.float32)
    key = ("golden", c["op"], o)
    worst[key] = max(worst[key], ratio(got, G.arr(c["output"])))
man, arrs = G.mid()
for c in man["cases"]:
    o = max(c["order"][:c["dim"]]) if isinstance(c["order"], list) else c["order"]
    ins = {k: np.asarray(arrs[v], np.float32) for k, v in c["inputs"].items()}
    b, od, e = c["bound"], c["order"], c["extrapolate"]
    op = c["op"]
    if op == "pull": got = hip.grid_pull(ins["inp"], ins["grid"], b, od, e)
    elif op == "grad": got = hip.grid_grad(ins["inp"], ins["grid"], b, od, e)
    elif op == "push": got = hip.grid_push(ins["inp"], ins["grid"], c["shape"], b, od, e)
    elif op == "count": got = hip.grid_count(ins["grid"], c["shape"], b, od, e)
    else: continue
    key = ("mid", op, o)
    worst[key] = max(worst[key], ratio(got, np.asarray(arrs[c["output"]], np.float64)))
for c in G.manifest()["prefilter"]:
    x = torch.from_numpy(G.arr(c["inp"])).to(DEV, torch.float32)
    fn = interpol.spline_coeff if c["fn"] == "spline_coeff" else interpol.spline_coeff_nd
    got = fn(x, interpolation=c["order"], bound=c["bound"], dim=c["dim"])
    o = max(c["order"]) if isinstance(c["order"], list) else c["order"]
    key = ("prefilter", c["fn"], o)
    worst[key] = max(worst[key], G.rel_err(got.cpu().numpy(), G.arr(c["out"])) / 1e-5)
for c in G.manifest()["backward"]:
    kw = dict(interpolation=c["interpolation"], bound=c["bound"], extrapolate=c["extrapolate"])
    grid = torch.from_numpy(G.arr(c["grid"])).to(DEV, torch.float32).requires_grad_(True)
    gout = torch.from_numpy(G.arr(c["gout"])).to(DEV, torch.float32)
    inp = None
    if c["fn"] == "grid_count":
        out = interpol.grid_count(grid, c["shape"], **kw)
    else:
        inp = torch.from_numpy(G.arr(c["inp"])).to(DEV, torch.float32).requires_grad_(True)
        out = interpol.grid_push(inp, grid, c["shape"], **kw) if c["fn"] == "grid_push" else getattr(interpol, c["fn"])(inp, grid, **kw)
    out.backward(gout)
    it = c["interpolation"]; o = max(it) if isinstance(it, (list, tuple)) else it
    key = ("backward", c["fn"], o)
    r = G.rel_err(out.detach().cpu().numpy(), G.arr(c["out"])) / 1e-5
    if inp is not None:
        r = max(r, G.rel_err(inp.grad.cpu().numpy(), G.arr(c["grad_inp"])) / 1e-5)
    rg = G.rel_err(grid.grad.cpu().numpy(), G.arr(c["grad_grid"])) / 1e-5
    worst[key] = max(worst[key], r)
    worst[("backward_grid", c["fn"], o)] = max(worst[("backward_grid", c["fn"], o)], rg)
print("# worst |err| in units of the 1e-5 tolerance (<= 1 passes); family, operator, spline order")
for k in sorted(worst, key=lambda k: (k[0], k[1], str(k[2]))):
    print("%-14s %-16s order %-3s %8.3f%s" % (k[0], k[1], k[2], worst[k], "   <-- above" if worst[k] > 1 else ""))
