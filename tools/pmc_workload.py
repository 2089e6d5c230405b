#!/usr/bin/env python
"""Workload for the PMC passes: a known-byte-count copy (calibration of FETCH_SIZE /
WRITE_SIZE in this access pattern, see MI355X_MICROARCH.md sec. HBM) followed by the
hot-path kernels at BASELINE config 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
torch.cuda.synchronize()
for _ in range(3):
    y = inp.clone()                      # 537 MB read + 537 MB written (calibration)
ops = os.environ.get("PMC_OPS", "pull,push").split(",")
for _ in range(3):
    if "pull" in ops:
        a = interpol.grid_pull(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
    if "push" in ops:
        b = interpol.grid_push(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
torch.cuda.synchronize()
