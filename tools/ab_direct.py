#!/usr/bin/env python
"""The single-pass small-box tiles of the pull (csrc/pull_direct.hip, opt-in: INTERPOL_FLAG_SMALL_TILES) against the class-sorted tiles alone
at config 2 (4 x 2 x 256^3 cubic dct2): identity, smooth field, i.i.d. noise; with the phase ablations (debug bits 1: no staging,
2: no taps).  ms per call, median of 5 x 4 calls."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
AUTO, SMALL = _hip.FLAG_AUTO_SCATTER, 1 << 25


def timeit(fn, reps=5, inner=4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]


order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cases = {"identity": bench.make_inputs(4, 2, 256, 0.0, dev, 1234)[1], "smooth": bench.smooth_grid(4, 256, 2.0, dev, 7)}
for s in (0.25, 0.5, 1.0, 2.0):
    cases["sigma_%g" % s] = bench.make_inputs(4, 2, 256, s, dev, 1234)[1]
inp = torch.randn([4, 2, 256, 256, 256], device=dev)
for name, grid in cases.items():
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [order] * 3, 1, flags=fl)
    ref = pf(_hip.FLAG_NO_FASTPATH)
    res = {"small_tiles": round(timeit(lambda: pf(AUTO | SMALL)), 3), "default": round(timeit(pf), 3),
           "err_vs_generic": "%.1e" % float((pf(AUTO | SMALL) - ref).abs().max() / ref.abs().max())}
    if name in ("identity", "smooth"):
        res["no_staging"] = round(timeit(lambda: pf(AUTO | SMALL | (1 << 8))), 3)
        res["no_taps"] = round(timeit(lambda: pf(AUTO | SMALL | (2 << 8))), 3)
        res["neither"] = round(timeit(lambda: pf(AUTO | SMALL | (3 << 8))), 3)
    print(name, json.dumps(res), flush=True)
