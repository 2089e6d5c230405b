#!/usr/bin/env python
"""The expanding push (target with >= 8 voxels per sample: interpol_push_bricks behind the API) and push + count at sizes with many bricks,
against the atomics-only kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(14)
NF = _hip.FLAG_NO_FASTPATH
bad = 0
for sshape, tshape in (((72, 64, 80), (160, 150, 170)), ((40, 48, 44), (200, 190, 180))):
    zoom = [(t - 1) / (s - 1) for s, t in zip(sshape, tshape)]
    ident = interpol.identity_grid(sshape)[None] * torch.tensor(zoom)
    for order in (1, 2, 3):
        for C in (1, 3):
            for sigma in (0.3, 3.0):
                for bound, ex in (("dct2", True), ("zero", False), ("dft", True)):
                    src = torch.randn([2, C, *sshape], generator=gen).to(dev)
                    grid = (ident + sigma * torch.randn([2, *sshape, 3], generator=gen)).contiguous().to(dev)
                    got = interpol.grid_push(src, grid, shape=list(tshape), interpolation=order, bound=bound, extrapolate=ex)
                    cnt = interpol.grid_count(grid, shape=list(tshape), interpolation=order, bound=bound, extrapolate=ex)
                    from interpol.codes import bound_to_code
                    b = [bound_to_code(bound)] * 3
                    ref = _hip.scatter("push", src, grid, list(tshape), b, [order] * 3, int(ex), flags=NF)
                    refc = _hip.scatter("count", None, grid, list(tshape), b, [order] * 3, int(ex), flags=NF)
                    for name, a, r in (("push", got, ref), ("count", cnt.reshape(refc.shape), refc)):
                        e = float((a - r).abs().max() / max(float(r.abs().max()), 1e-30))
                        if not e < 1e-5:
                            bad += 1
                            print("BAD", name, sshape, tshape, order, C, sigma, bound, e, flush=True)
    print("done", sshape, tshape, "bad so far", bad, flush=True)
print("sweep4: bad =", bad, flush=True)
