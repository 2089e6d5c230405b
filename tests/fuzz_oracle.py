#!/usr/bin/env python
"""Randomised parity run against the oracle (not collected by pytest; run by hand on a GPU box: python tests/fuzz_oracle.py
[n_cases] [seed]).  Operators through the C-ABI with the default routing (tile kernels, hand-back, ...) vs the fp64 oracle
(oracle/: test infrastructure, the CPU restatement of nd.py / pushpull.py) on random problems of a size the tile kernels
accept: dims 1-3 (1-D: the scatter tiles of push1d.hip), mixed orders and bounds, the three extrapolation modes, identity / noisy / zoomed / rough lattices,
broadcast batches.  Tolerance: the parity bar of tests/golden_util.py (fp32: 1e-5 relative + 1e-5 of the largest value,
looser for scatters that pile thousands of samples on a voxel)."""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, interpol
from interpol import ops
from oracle import oracle
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); gen = torch.Generator().manual_seed(seed)
oracle.set_threads(min(32, os.cpu_count() or 1))
def rel(a, r):
    r = np.asarray(r, dtype=np.float64); a = a.detach().double().cpu().numpy()
    return float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-20))
bad = 0
for case in range(n_cases):
    dim = rnd.choice([1, 2, 3, 3, 3]); B = rnd.choice([1, 2]); C = rnd.choice([1, 2, 3])
    ishape = [rnd.randint(17, 40) for _ in range(3)] if dim == 3 else ([rnd.randint(40, 150) for _ in range(2)] if dim == 2 else [rnd.randint(300, 7000)])
    oshape = [rnd.randint(17, 40) for _ in range(3)] if dim == 3 else ([rnd.randint(65, 150) for _ in range(2)] if dim == 2 else [rnd.randint(4096, 9000)])
    order = [rnd.choice([0, 1, 2, 3, 3, 4, 5, 6, 7])] * dim if rnd.random() < 0.6 else [rnd.choice([1, 2, 3]) for _ in range(dim)]
    bound = [rnd.randrange(7) for _ in range(dim)] if rnd.random() < 0.5 else [rnd.randrange(7)] * dim
    ex = rnd.choice([0, 1, 1, 2])
    kind = rnd.choice(["identity", "noise", "zoom", "rough"])
    scale = (torch.tensor(ishape, dtype=torch.float32) - 1) / (torch.tensor(oshape, dtype=torch.float32) - 1)
    g0 = interpol.identity_grid(oshape) * scale
    if kind == "noise": g0 = g0 + rnd.choice([0.5, 2.0]) * torch.randn(g0.shape, generator=gen)
    if kind == "zoom": g0 = (g0 - g0.reshape(-1, dim).mean(0)) * rnd.choice([1.7, 2.5]) + g0.reshape(-1, dim).mean(0)
    if kind == "rough": g0 = g0 + 5.0 * torch.randn(g0.shape, generator=gen)
    gb = rnd.choice([1, B])
    grid = (g0[None] + 0.05 * torch.randn([gb, *oshape, dim], generator=gen)).contiguous()
    # keep the coordinates away from the thresholds of the extrapolation mask (nd.py:10-27): a float32 coordinate ON the float32
    # threshold is outside for the reference's float32 compare and inside for the oracle's float64 one
    for d in range(dim):
        for thr in (-0.05, -0.55, ishape[d] - 1 + 0.05, ishape[d] - 1 + 0.55):
            near = (grid[..., d] - thr).abs() < 1e-3
            grid[..., d] = torch.where(near, grid[..., d] + 0.01, grid[..., d])
    vol = torch.randn([B, C, *ishape], generator=gen); src = torch.randn([B, C, *oshape], generator=gen); gvo = torch.randn([B, C, *ishape], generator=gen)
    hi = max(order); tol = 3e-5 if hi < 4 else (1e-4 if hi < 7 else 5e-4); pile = 1e-3
    G, V, S, GV = grid.to(dev), vol.to(dev), src.to(dev), gvo.to(dev)
    gn, vn, sn, gvn = grid.double().numpy(), vol.double().numpy(), src.double().numpy(), gvo.double().numpy()
    if gb != B: gn = np.broadcast_to(gn, (B,) + gn.shape[1:]).copy()
    checks = []
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]): continue
    if os.environ.get("FUZZ_ONLY"):
        from interpol import _hip
        for fl, name in ((0, "default"), (_hip.FLAG_NO_FASTPATH, "generic"), (256 << 8, "no hand-back")):
            print(name, "pull vs oracle %.2e" % rel(_hip.gather("pull", V, G, bound, order, ex, flags=fl), oracle.grid_pull(vn, gn, bound, order, ex)),
                  "f64 generic vs oracle %.2e" % rel(_hip.gather("pull", V.double(), G.double(), bound, order, ex, flags=fl), oracle.grid_pull(vn, gn, bound, order, ex)))
    try:
        checks.append(("pull", rel(ops.grid_pull(V, G, bound, order, ex), oracle.grid_pull(vn, gn, bound, order, ex)), tol))
        checks.append(("grad", rel(ops.grid_grad(V, G, bound, order, ex), oracle.grid_grad(vn, gn, bound, order, ex)), 3 * tol))
        checks.append(("push", rel(ops.grid_push(S, G, ishape, bound, order, ex), oracle.grid_push(sn, gn, ishape, bound, order, ex)), pile))
        checks.append(("count", rel(ops.grid_count(G, ishape, bound, order, ex), oracle.grid_count(gn, ishape, bound, order, ex)), pile))
        gi, gg = ops.grid_pull_backward(S, V, G, bound, order, ex, need_inp=True, need_grid=True)
        wi, wg = oracle.grid_pull_backward(sn, vn, gn, bound, order, ex)
        # (operator level: one gradient slab per batch item, the autograd layer sums those of a broadcast grid)
        checks += [("pull bwd inp", rel(gi, wi), pile), ("pull bwd grid", rel(gg, wg), 10 * tol)]
        gi, gg = ops.grid_push_backward(GV, S, G, bound, order, ex, need_inp=True, need_grid=True)
        wi, wg = oracle.grid_push_backward(gvn, sn, gn, bound, order, ex)
        checks += [("push bwd inp", rel(gi, wi), tol), ("push bwd grid", rel(gg, wg), 10 * tol)]
    except Exception as e:
        print("EXCEPTION", case, dim, B, C, ishape, oshape, order, bound, ex, kind, gb, repr(e)); bad += 1; continue
    fails = [(k, "%.1e" % v) for k, v, t in checks if not v <= t]
    if fails:
        bad += 1; print("MISMATCH", case, "dim", dim, "B", B, "C", C, ishape, oshape, "order", order, "bound", bound, "ex", ex, kind, "grid batch", gb, fails)
print("fuzz vs oracle: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
