#!/usr/bin/env python
"""Workload for tools/kstats.sh: BASELINE config 3 (8 x 1 x 192^3 fp32, order 5, dft, sigma = 2): pull, grid_grad, pull_backward with both gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(8, 1, 192, 192, 192, generator=g, device=dev)
grid = torch.randn([8, 192, 192, 192, 3], generator=g, device=dev).mul_(2.0) + interpol.identity_grid([192] * 3, device=dev)
src = torch.randn_like(x)
b, o = [6] * 3, [5] * 3
for _ in range(5):
    _hip.gather("pull", x, grid, b, o, 1)
    _hip.gather("grad", x, grid, b, o, 1)
    _hip.pull_backward(src, x, grid, b, o, 1, True, True)
torch.cuda.synchronize()
