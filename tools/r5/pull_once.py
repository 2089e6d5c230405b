#!/usr/bin/env python
"""A few pulls at config 2 (for rocprofv3 --kernel-trace): argv = sigma [flags]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
for _ in range(12):
    a = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=flags)
torch.cuda.synchronize()
