#!/usr/bin/env python
"""Debug aid for the mixed-order tiles: the failing case of test_sorted_tiles_mixed_orders_against_the_oracle, the samples that differ from the
generic kernel with their coordinates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch
from interpol import _hip
DEV = torch.device("cuda", 0)
orders = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "112")]
gen = torch.Generator().manual_seed(100 * orders[0] + 10 * orders[1] + orders[2])
ishape, oshape = (50, 37, 41), (37, 45, 50)
ident = torch.stack(torch.meshgrid(*[torch.linspace(0, n - 1, m) for n, m in zip(ishape, oshape)], indexing="ij"), -1)[None]
for case, (bounds, ex, C, sigma) in enumerate((([3, 3, 3], 1, 2, 2.0), ([2, 5, 0], 0, 1, 0.5), ([6, 1, 4], 2, 3, 3.0), ([4, 0, 6], 1, 2, 0.5))):
    vol = torch.randn([2, C, *ishape], generator=gen)
    src = torch.randn([2, C, *oshape], generator=gen)
    grid = (ident + sigma * torch.randn([2, *oshape, 3], generator=gen)).contiguous()
    grid[0, 0, 0, 0] = -3.0 * torch.tensor(ishape)
    grid[1, 1, 2, 3] = 3.0 * torch.tensor(ishape) + 0.25
    vd, gd = vol.to(DEV), grid.to(DEV)
    fast = _hip.gather("pull", vd, gd, bounds, orders, ex)
    slow = _hip.gather("pull", vd, gd, bounds, orders, ex, flags=_hip.FLAG_NO_FASTPATH)
    old = _hip.gather("pull", vd, gd, bounds, orders, ex, flags=16 << 8)
    d = (fast - slow).abs()
    bad = (d > 1e-4 * slow.abs().max()).nonzero()
    print("case", case, bounds, ex, C, sigma, "max diff", float(d.max()), "bad", len(bad), "round-1 tiles vs generic", float((old - slow).abs().max()))
    for idx in bad[:8].tolist():
        b, c, x, y, z = idx
        print("  sample", idx, "coord", grid[b, x, y, z].tolist(), "fast", float(fast[b, c, x, y, z]), "generic", float(slow[b, c, x, y, z]))
    if "--oracle" in sys.argv:
        import numpy as np
        from oracle import oracle
        oracle.set_threads(os.cpu_count() or 8)
        want = torch.from_numpy(np.asarray(oracle.grid_pull(vol.double(), grid.double(), bounds, orders, ex))).float()
        d = (slow.cpu() - want).abs()
        bad = (d > 1e-4 * want.abs().max()).nonzero()
        print("  generic vs oracle: max diff", float(d.max()), "bad", len(bad))
        for idx in bad[:8].tolist():
            b, c, x, y, z = idx
            print("   sample", idx, "coord", [repr(float(v)) for v in grid[b, x, y, z]], "generic", float(slow[b, c, x, y, z]), "oracle", float(want[b, c, x, y, z]))
