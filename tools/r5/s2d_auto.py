#!/usr/bin/env python
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 32, 3, 1024
x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
ident = interpol.identity_grid([n, n], device=dev)[None]
bc, o = [2, 5], [2, 3]
for sigma in (2.0, 8.0):
    grid = (ident + sigma * torch.randn(B, n, n, 2, generator=g, device=dev)).contiguous()
    for _ in range(6):
        _hip.scatter("push", x, grid, None, bc, o, 1)
    torch.cuda.synchronize()
