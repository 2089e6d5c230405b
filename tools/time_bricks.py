#!/usr/bin/env python
"""The brick organisations at config 2 (4 x 2 x 256^3 cubic dct2) over sigma: owner-computes push, bricks-of-the-image pull and grid
gradient (INTERPOL_FLAG_BINNED_SCATTER) next to the routed defaults.  ms per call, median of 5 x 4 back-to-back calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)


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


B = _hip.FLAG_BINNED_SCATTER
for s in [float(a) for a in sys.argv[1:]] or (2.0, 0.0, 6.0):
    inp, grid = bench.make_inputs(4, 2, 256, s, dev, 1234)
    gout = torch.randn_like(inp)
    f = lambda fl=0: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=fl)
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    gf = lambda fl=0: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, False, True, flags=fl)[1]
    pb = lambda: _hip.push_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True)
    sg = lambda fl=0: _hip.gather("grad", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    ref_p, ref_g = pf(_hip.FLAG_NO_FASTPATH), gf(_hip.FLAG_NO_FASTPATH)
    res = {"push_owner": timeit(lambda: f(B)), "push_default": timeit(f),
           "pull_bricks": timeit(lambda: pf(B)), "pull_default": timeit(pf),
           "gradgrid_bricks": timeit(lambda: gf(B)), "gradgrid_default": timeit(gf),
           "push_backward_both": timeit(pb), "grad_bricks": timeit(lambda: sg(B)), "grad_default": timeit(sg), "grad_tiles": timeit(lambda: sg(_hip.FLAG_FORCE_TILED))}
    ref_s = sg(_hip.FLAG_NO_FASTPATH)
    gerr = float((sg(B) - ref_s).abs().max() / ref_s.abs().max())
    print("sigma", s, {k: round(v, 3) for k, v in res.items()},
          "err pull %.1e gradgrid %.1e grad %.1e" % (float((pf(B) - ref_p).abs().max() / ref_p.abs().max()), float((gf(B) - ref_g).abs().max() / ref_g.abs().max()), gerr), flush=True)
