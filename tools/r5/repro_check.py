#!/usr/bin/env python
"""Is the owner-computes push bit-reproducible?  N runs against the first, differing voxels listed."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
ref = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
gen = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_NO_FASTPATH)
print("vs generic: max abs", float((ref - gen).abs().max()), "max", float(gen.abs().max()))
for i in range(n):
    b = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
    d = (b != ref)
    nd = int(d.sum())
    if nd:
        idx = d.nonzero()[:8]
        print("run", i, "n_diff", nd, [(tuple(int(t) for t in ix), float(ref[tuple(ix)]), float(b[tuple(ix)]), float(gen[tuple(ix)])) for ix in idx])
    else:
        print("run", i, "identical")
