#!/usr/bin/env python
"""Differential fuzz: every operator through the default routing (tile kernels, hand-back, ...) vs the generic kernels
(INTERPOL_FLAG_NO_FASTPATH) on random problems: dims 2-3, mixed orders and bounds, the three extrapolation modes, ragged
shapes, 1-5 channels, f32 / bf16 / f16 storage, deformations from the identity to zooms, strides, smooth and rough fields.
usage: tools/fuzz_fastpaths.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
GEN = _hip.FLAG_NO_FASTPATH
def rel(a, r):
    a, r = a.float(), r.float()
    return float((a - r).abs().max() / r.abs().max().clamp_min(1e-20))
bad = 0
for case in range(n_cases):
    dim = rnd.choice([2, 3, 3])
    dt = rnd.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
    B, C = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3, 5])
    if dim == 3:
        ishape = [rnd.randint(20, 90) for _ in range(3)]; oshape = [rnd.randint(17, 70) for _ in range(3)]
    else:
        ishape = [rnd.randint(40, 400) for _ in range(2)]; oshape = [rnd.randint(33, 300) for _ in range(2)]
    same = rnd.random() < 0.5
    order = [rnd.choice([1, 2, 3, 3, 5])] * dim if same else [rnd.choice([1, 2, 3]) for _ in range(dim)]
    bound = [rnd.randrange(7)] * dim if rnd.random() < 0.5 else [rnd.randrange(7) for _ in range(dim)]
    ex = rnd.choice([1, 1, 0, 2])
    kind = rnd.choice(["identity", "noise", "zoom", "stride", "smooth", "rough"])
    ident = interpol.identity_grid(oshape)
    scale = (torch.tensor(ishape, dtype=torch.float32) - 1) / (torch.tensor(oshape, dtype=torch.float32) - 1)
    g0 = ident * scale
    if kind == "noise": g0 = g0 + rnd.choice([0.3, 1.0, 2.0]) * torch.randn(g0.shape, generator=gen)
    elif kind == "zoom": g0 = (g0 - g0.mean()) * rnd.choice([1.5, 2.0, 2.7]) + g0.mean()
    elif kind == "stride": g0 = ident * rnd.choice([1.9, 2.2, 3.0])
    elif kind == "smooth":
        ctrl = torch.randn([1, dim] + [5] * dim, generator=gen) * rnd.choice([2.0, 6.0])
        g0 = g0 + torch.nn.functional.interpolate(ctrl, size=oshape, mode="bicubic" if dim == 2 else "trilinear", align_corners=True)[0].movedim(0, -1)
    elif kind == "rough": g0 = g0 + rnd.choice([4.0, 7.0]) * torch.randn(g0.shape, generator=gen)
    grid = (g0[None] + 0.02 * torch.randn([B, *oshape, dim], generator=gen)).contiguous().to(dev)
    vol = torch.randn([B, C, *ishape], generator=gen).to(dev).to(dt)
    src = torch.randn([B, C, *oshape], generator=gen).to(dev).to(dt)
    gvo = torch.randn([B, C, *ishape], generator=gen).to(dev).to(dt)
    lowp = dt != torch.float32
    hi = max(order)
    tol = 2e-2 if lowp else (2e-5 if hi < 5 else 2e-4)
    checks = []
    try:
        checks.append(("pull", rel(_hip.gather("pull", vol, grid, bound, order, ex), _hip.gather("pull", vol, grid, bound, order, ex, flags=GEN))))
        checks.append(("grad", rel(_hip.gather("grad", vol, grid, bound, order, ex), _hip.gather("grad", vol, grid, bound, order, ex, flags=GEN))))
        checks.append(("push", rel(_hip.scatter("push", src, grid, ishape, bound, order, ex), _hip.scatter("push", src, grid, ishape, bound, order, ex, flags=GEN))))
        checks.append(("count", rel(_hip.scatter("count", None, grid, ishape, bound, order, ex), _hip.scatter("count", None, grid, ishape, bound, order, ex, flags=GEN))))
        for need in ((True, True), (False, True)):
            a = _hip.pull_backward(src, vol, grid, bound, order, ex, *need); r = _hip.pull_backward(src, vol, grid, bound, order, ex, *need, flags=GEN)
            checks += [("pullbwd%d%d_%d" % (need + (i,)), rel(x, y)) for i, (x, y) in enumerate(zip(a, r)) if x is not None]
            a = _hip.push_backward(gvo, src, grid, bound, order, ex, *need); r = _hip.push_backward(gvo, src, grid, bound, order, ex, *need, flags=GEN)
            checks += [("pushbwd%d%d_%d" % (need + (i,)), rel(x, y)) for i, (x, y) in enumerate(zip(a, r)) if x is not None]
    except Exception as e:
        print("EXCEPTION", case, dim, dt, B, C, ishape, oshape, order, bound, ex, kind, repr(e)); bad += 1; continue
    # scatters of samples that pile up on a few voxels (zooms beyond the volume under a clamping bound: counts of 10^4 - 10^5)
    # carry float32 accumulation error in BOTH paths (3e-4 of the maximum against the f64 kernels, the reference's
    # scatter_add_ likewise): the two float32 results may differ by as much
    def tol_of(name):
        pile = kind in ("zoom", "stride", "rough") and name[:4] in ("push", "coun", "pull") and name not in ("pull",) and not lowp
        return 5e-4 if pile and not name.endswith("_1") else tol
    fails = [(k, "%.1e" % v) for k, v in checks if not v <= tol_of(k)]
    if fails:
        bad += 1
        print("MISMATCH case", case, "dim", dim, dt, "B", B, "C", C, ishape, oshape, "order", order, "bound", bound, "ex", ex, kind, fails)
print("fuzz: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
