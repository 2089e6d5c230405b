#!/usr/bin/env python
"""Small workload for rocprofv3 --kernel-trace --stats: config-2 operators a few times each.
usage: prof_ops.py <sigma> <op>[,<op>...]   ops: pull, push, push_binned, push_sorted, pull_bwd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ops = (sys.argv[2] if len(sys.argv) > 2 else "pull,push").split(",")
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
kw = dict(interpolation=3, bound="dct2", extrapolate=True)
for _ in range(4):
    for op in ops:
        if op == "pull": interpol.grid_pull(inp, grid, **kw)
        elif op == "push": interpol.grid_push(inp, grid, **kw)
        elif op in ("push_binned", "push_owner"):
            backend.rough_deformations = True
            interpol.grid_push(inp, grid, **kw)
            backend.rough_deformations = None
        elif op == "push_sorted":
            _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=128 << 8)
        elif op == "count": interpol.grid_count(grid, **kw)
torch.cuda.synchronize()
