#!/usr/bin/env python
"""Ablation timings of the 2-D push tile kernel at config 5 (dbg bits: 1 no taps, 2 no flush)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
B, C, n = 32, 3, 1024
gen = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, C, n, n, generator=gen, device=dev).to(torch.bfloat16)
for sigma in (2.0, 0.0):
    gr = torch.randn([B, n, n, 2], generator=gen, device=dev).mul_(sigma) + interpol.identity_grid([n, n], device=dev)
    out = {}
    op = sys.argv[1] if len(sys.argv) > 1 else "push"
    for dbg in ((0, 1, 2, 3) if op == "push" else (0, 1, 2, 4, 8, 3, 7, 15)):
        f = (lambda: _hip.scatter("push", x, gr, None, [2, 5], [2, 3], 1, flags=dbg << 8)) if op == "push" else (lambda: _hip.gather("pull", x, gr, [2, 5], [2, 3], 1, flags=dbg << 8))
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        out[dbg] = round(a.elapsed_time(b) / 10, 3)
    print(sigma, json.dumps(out))
