#!/usr/bin/env python
"""Default-flag scatter (and, since round 4, gather) on the deformations that used to be cliffs: i.i.d. noise, a folding smooth field, strides, zooms.
Prints ms per call: default routing (probe: tiles or owner-computes), tiles only, owner-computes only, and the largest difference of the
default result to the generic kernels."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)
def timeit(fn, reps=3, inner=4):
    """median over `reps` of the time per call of `inner` back-to-back calls (a single call after a synchronisation runs
    at idle clocks: +15-20 % on this chip)"""
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
n = 256
inp, ident = bench.make_inputs(4, 2, n, 0.0, dev, 1234)
g = torch.Generator(device=dev).manual_seed(7)
cases = {}
for s in (3.0, 6.0):
    cases["iid_sigma_%g" % s] = ident + s * torch.randn(ident.shape, generator=g, device=dev)
ctrl = torch.randn(4, 3, 12, 12, 12, generator=g, device=dev) * 8.0
disp = interpol.resize(ctrl, shape=[n] * 3, anchor="e", interpolation=3, bound="dct2", prefilter=True)
cases["smooth_amp_8"] = ident + disp.permute(0, 2, 3, 4, 1)
cases["zoom_2"] = (ident - (n - 1) / 2) * 2.0 + (n - 1) / 2
cases["zoom_1.5"] = (ident - (n - 1) / 2) * 1.5 + (n - 1) / 2
cases["stride_2_inside"] = ident * 0.5 + 20.0
del ctrl, disp
for name, grid in cases.items():
    grid = grid.contiguous()
    res = {}
    f = lambda fl=0: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=fl)
    backend.rough_deformations = None
    res["default"] = round(timeit(f), 2)
    backend.rough_deformations = False
    res["tiles"] = round(timeit(f), 2)
    backend.rough_deformations = None
    res["owner"] = round(timeit(lambda: f(_hip.FLAG_BINNED_SCATTER)), 2)
    a, r = f(), f(_hip.FLAG_NO_FASTPATH)
    res["rel_err_vs_generic"] = "%.1e" % float((a - r).abs().max() / r.abs().max())
    del a, r
    # the grid gradient of the pull (backward w.r.t. the grid): routed / sample tiles only / bricks only, error against the generic kernel
    gout = torch.randn_like(inp)
    gf = lambda fl=0: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, False, True, flags=fl)[1]
    res["gradgrid_default"] = round(timeit(gf), 2)
    backend.rough_deformations = False
    res["gradgrid_tiles"] = round(timeit(gf), 2)
    backend.rough_deformations = None
    res["gradgrid_bricks"] = round(timeit(lambda: gf(_hip.FLAG_BINNED_SCATTER)), 2)
    a, a2, r = gf(), gf(_hip.FLAG_BINNED_SCATTER), gf(_hip.FLAG_NO_FASTPATH)
    res["gradgrid_rel_err_vs_generic"] = "%.1e / bricks %.1e" % (float((a - r).abs().max() / r.abs().max()), float((a2 - r).abs().max() / r.abs().max()))
    del a, a2, r, gout
    if name in ("smooth_amp_8", "iid_sigma_6"):
        gout = torch.randn_like(inp)
        res["pull_backward_both"] = round(timeit(lambda: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True)), 2)
        backend.rough_deformations = False
        res["pull_backward_both_tiles"] = round(timeit(lambda: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True)), 2)
        backend.rough_deformations = None
        ga = _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True)
        gr_ = _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True, flags=_hip.FLAG_NO_FASTPATH)
        res["bwd_rel_err"] = "%.1e" % max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(ga, gr_))
        del gout, ga, gr_
    # the pull of the same field: routed (the tiles leave the rough tiles to bricks of the image), tiles only, bricks only
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    res["pull_default"] = round(timeit(pf), 2)
    res["pull_tiles"] = round(timeit(lambda: pf(_hip.FLAG_FORCE_TILED)), 2)
    res["pull_bricks"] = round(timeit(lambda: pf(_hip.FLAG_BINNED_SCATTER)), 2)
    a, r = pf(), pf(_hip.FLAG_NO_FASTPATH)
    res["pull_rel_err_vs_generic"] = "%.1e" % float((a - r).abs().max() / r.abs().max())
    del a, r
    print(name, json.dumps(res), flush=True)
