#!/usr/bin/env python
"""Differential fuzz of the target-stationary scatters against the generic kernels: interpol_push_bricks (expanding fields,
csrc/push_bricks.hip) and INTERPOL_FLAG_BINNED_SCATTER (owner-computes, csrc/push_owner.hip) -- random shapes, orders 0-3,
every bound, extrapolation modes, 1-3 channels, count channel, shared targets, expansions 1-5x, noise up to 8 voxels.
usage: tools/fuzz_scatter_variants.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); gen = torch.Generator().manual_seed(seed)
GEN = _hip.FLAG_NO_FASTPATH
def rel(a, r): return float((a - r).abs().max() / r.abs().max().clamp_min(1e-20))
bad = 0
for case in range(n_cases):
    variant = rnd.choice(["bricks", "binned"])
    B, C = rnd.choice([1, 2, 4]), rnd.choice([1, 2, 3])
    sshape = [rnd.randint(9, 40) for _ in range(3)] if variant == "bricks" else [rnd.randint(12, 56) for _ in range(3)]    # (>= 32: the end bricks fold)
    expand = rnd.choice([1.0, 2.0, 3.3, 5.0]) if variant == "bricks" else rnd.choice([1.0, 1.3, 0.45])     # 0.45: half of the samples beyond the lattice
    tshape = [max(4, int(s * expand) + rnd.randint(-2, 3)) for s in sshape]
    order = [rnd.choice([0, 1, 2, 3, 3])] * 3; bound = [rnd.randrange(7)] * 3 if rnd.random() < 0.6 else [rnd.randrange(7) for _ in range(3)]
    ex = rnd.choice([0, 1, 1, 2]); wc = rnd.random() < 0.4 and C < 3; shared = rnd.random() < 0.4
    sigma = rnd.choice([0.0, 0.5, 2.0]) if variant == "bricks" else rnd.choice([0.5, 3.0, 8.0])
    scale = (torch.tensor(tshape, dtype=torch.float32) - 1) / (torch.tensor(sshape, dtype=torch.float32) - 1)
    if expand < 1: scale = torch.ones(3)                             # (the sample grid keeps its pitch: it overhangs the lattice)
    grid = (interpol.identity_grid(sshape) * scale)[None] + sigma * torch.randn([B, *sshape, 3], generator=gen)
    grid = grid.contiguous().to(dev); src = torch.randn([B, C, *sshape], generator=gen).to(dev)
    try:
        ref = _hip.scatter("push", src, grid, tshape, bound, order, ex, flags=GEN, shared=shared, with_count=wc)
        if variant == "bricks":
            got = _hip.push_bricks(src, grid, tshape, bound, order, ex, shared=shared, with_count=wc)
        else:
            got = _hip.scatter("push", src, grid, tshape, bound, order, ex, flags=_hip.FLAG_BINNED_SCATTER, shared=shared, with_count=wc)
        e = rel(got, ref)
        if variant == "binned" and not wc:
            e = max(e, rel(_hip.scatter("count", None, grid, tshape, bound, order, ex, flags=_hip.FLAG_BINNED_SCATTER, shared=shared),
                           _hip.scatter("count", None, grid, tshape, bound, order, ex, flags=GEN, shared=shared)))
    except Exception as ex_:
        print("EXCEPTION", case, variant, B, C, sshape, tshape, order, bound, ex, wc, shared, sigma, repr(ex_)); bad += 1; continue
    tol = 2e-4
    if expand < 1 and not e <= tol:
        # overhanging sample grids pile thousands of float additions onto the faces of the lattice (replicate: onto single points):
        # the fp32 generic kernel is itself noisy there -- both are judged against the fp64 generic kernel
        ref64 = _hip.scatter("push", src.double(), grid.double(), tshape, bound, order, ex, flags=GEN, shared=shared, with_count=wc)
        e, tol = rel(got.double(), ref64), max(tol, 3.0 * rel(ref.double(), ref64))
    if not e <= tol:
        bad += 1; print("MISMATCH", case, variant, "B", B, "C", C, sshape, "->", tshape, "order", order, "bound", bound, "ex", ex, "count", wc, "shared", shared, "sigma", sigma, "err %.1e" % e)
print("fuzz scatter variants: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
