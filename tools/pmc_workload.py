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
if any(o.endswith("2d") for o in os.environ.get("PMC_OPS", "").split(",")):
    # config 5's shape (one GPU's share): 32 x 3 x 1024^2 bf16, orders [2, 3], bounds [dct1, dst2]
    gen = torch.Generator(device=dev).manual_seed(5)
    x2 = torch.randn(32, 3, 1024, 1024, generator=gen, device=dev).to(torch.bfloat16)
    g2 = torch.randn([32, 1024, 1024, 2], generator=gen, device=dev).mul_(sigma) + interpol.identity_grid([1024, 1024], device=dev)
    kw2 = dict(interpolation=[2, 3], bound=["dct1", "dst2"], extrapolate=True)
    for _ in range(3):
        if "pull2d" in os.environ["PMC_OPS"]: interpol.grid_pull(x2, g2, **kw2)
        if "push2d" in os.environ["PMC_OPS"]: interpol.grid_push(x2, g2, **kw2)
    torch.cuda.synchronize()
torch.cuda.synchronize()
for _ in range(3):
    y = inp.clone()                      # 537 MB read + 537 MB written (calibration)
ops = os.environ.get("PMC_OPS", "pull,push").split(",")
for _ in range(3):
    if "pull" in ops:
        a = interpol.grid_pull(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
    if "pull4" in ops:                   # the four-pass tiles (pull_sorted), debug bit 4096
        from interpol import _hip
        a = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=4096 << 8)
    if "push" in ops:
        b = interpol.grid_push(inp, grid, interpolation=3, bound="dct2", extrapolate=True)
torch.cuda.synchronize()
