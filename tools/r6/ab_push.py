#!/usr/bin/env python
"""Owner-computes push at config 2: ONE launch over the interior bricks (round 6, default) against the eight colour launches of rounds
3-5 (debug bit 1024): ms per call and the largest difference between the two (and against the generic kernel on a smaller case).
argv: sigmas."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=9, inner=4):
    for _ in range(2):
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
COL = 1024 << 8
res = {}
# parity on a case the generic kernel finishes quickly: 2 x 2 x 96 x 112 x 104, three bounds
for bnd in (3, 1, 6, 0):
    inp, grid = bench.make_inputs(2, 2, 96, 3.0, dev, 7)
    ref = _hip.scatter("push", inp, grid, None, [bnd] * 3, [3] * 3, 1, flags=_hip.FLAG_NO_FASTPATH)
    new = _hip.scatter("push", inp, grid, None, [bnd] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
    old = _hip.scatter("push", inp, grid, None, [bnd] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | COL)
    acc = torch.ones_like(ref)
    _hip.scatter("push", inp, grid, None, [bnd] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | _hip.FLAG_ACCUMULATE, out=acc)
    m = ref.abs().max().item()
    res["bound%d" % bnd] = {"new_vs_generic": (new - ref).abs().max().item() / m, "old_vs_generic": (old - ref).abs().max().item() / m,
                           "accumulate_vs_generic": (acc - 1 - ref).abs().max().item() / m}
for sigma in [float(s) for s in sys.argv[1:]] or [2.0, 0.0]:
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    a = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
    b = _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | COL)
    res["diff_%g" % sigma] = (a - b).abs().max().item() / b.abs().max().item()
    del a, b
    res["one_launch_%g" % sigma] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)), 4)
    res["colours_%g" % sigma] = round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | COL)), 4)
    res["count_one_launch_%g" % sigma] = round(timeit(lambda: _hip.scatter("count", None, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)), 4)
    res["count_colours_%g" % sigma] = round(timeit(lambda: _hip.scatter("count", None, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | COL)), 4)
print(json.dumps(res, indent=1), flush=True)
