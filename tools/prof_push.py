#!/usr/bin/env python
"""Workload for rocprofv3 --kernel-trace --stats: grid_push / grid_pull at BASELINE config 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
for _ in range(4):
    b = interpol.grid_push(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
    a = interpol.grid_pull(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
torch.cuda.synchronize()
