#!/usr/bin/env python
"""Fast paths vs the generic kernels on ZOOMED lattices (grid = centre + s * (identity - centre)): a 16^3 sample tile
then spans (16 s)^3 lattice points; beyond s ~ 1.8 it no longer fits the LDS box and the tile kernels hand their tiles
back to the generic kernels (csrc/defer.hip).  Prints ms per call (hand-back on / off (dbg 256) / generic) and the
largest difference to the generic result."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench
dev = torch.device("cuda", 0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts)//2]
def rel(a, r):
    return float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-30))
NOHB, GEN = 256 << 8, _hip.FLAG_NO_FASTPATH
quick = "quick" in sys.argv
cases = [("3d_o3", 3, 4, 2, 128 if quick else 256, 3, torch.float32), ("3d_o5", 3, 2, 1, 96 if quick else 192, 5, torch.float32),
         ("3d_o1_c3", 3, 2, 3, 96 if quick else 192, 1, torch.float32), ("2d_o23_bf16", 2, 8, 3, 512 if quick else 1024, None, torch.bfloat16)]
for name, dim, B, C, n, order, dt in cases:
    g = torch.Generator(device=dev).manual_seed(7)
    inp = torch.randn([B, C] + [n] * dim, generator=g, device=dev).to(dt)
    gout = torch.randn([B, C] + [n] * dim, generator=g, device=dev).to(dt)
    grid0 = interpol.identity_grid([n] * dim, device=dev)[None].expand(B, *[n] * dim, dim).contiguous()
    o = [order] * dim if order is not None else [2, 3]
    bnd = [3] * dim
    for s in (1.0, 1.5, 2.0, 3.0):
        grid = (grid0 - (n - 1) / 2) * s + (n - 1) / 2
        res = {}
        ops = {
            "pull": lambda fl: _hip.gather("pull", inp, grid, bnd, o, 1, flags=fl),
            "grad": lambda fl: _hip.gather("grad", inp, grid, bnd, o, 1, flags=fl),
            "push": lambda fl: _hip.scatter("push", inp, grid, None, bnd, o, 1, flags=fl),
            "count": lambda fl: _hip.scatter("count", None, grid, [n] * dim, bnd, o, 1, flags=fl),
            "pullbwd": lambda fl: _hip.pull_backward(gout, inp, grid, bnd, o, 1, True, True, flags=fl),
            "pullbwd_grid": lambda fl: _hip.pull_backward(gout, inp, grid, bnd, o, 1, False, True, flags=fl)[1],
            "pushbwd": lambda fl: _hip.push_backward(gout, inp, grid, bnd, o, 1, True, True, flags=fl),
        }
        for op, fn in ops.items():
            a, r = fn(0), fn(GEN)
            if isinstance(a, tuple):
                err = max(rel(x, y) for x, y in zip(a, r))
            else:
                err = rel(a, r)
            del a, r
            res[op] = [round(timeit(lambda: fn(0)), 2), round(timeit(lambda: fn(NOHB)), 2) if s >= 2 else None, round(timeit(lambda: fn(GEN)), 2), "%.1e" % err]
        print("zoom", name, s, json.dumps(res), flush=True)
