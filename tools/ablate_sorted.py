#!/usr/bin/env python
"""Ablation timings of the class-sorted tile kernels (debug bits of interpol_problem.flags >> 8:
1 no staging / flush, 2 no tap loop, 4 flush without its global atomics, 16 flush every slot of the box, 64 staging loads from a 64 KiB window;
   op "pushs" = push_sorted, switch 128)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench
dev = torch.device("cuda", 0)
B, C, n = 4, 2, 256
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
ops = sys.argv[1:] or ["pull"]
for sigma in (2.0, 0.0):
    inp, grid = bench.make_inputs(B, C, n, sigma, dev, 1234)
    res = {}
    for name, flags in (("full", 0), ("no_stage", 1 << 8), ("no_taps", 2 << 8), ("neither", 3 << 8), ("stage_hits", 64 << 8), ("stage_hits_no_taps", 66 << 8)):
        if "pull" in ops:
            res["pull_" + name] = timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=flags))
        if "push" in ops:       # the routed push (push_tiled unless the sorted one is the default)
            res["push_" + name] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=flags))
        if "pushs" in ops and name in ("full", "no_stage", "no_taps", "neither"):      # push_sorted (switch 128): 1 = no flush
            res["pushs_" + name] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=flags | (128 << 8)))
            if name == "full":
                res["pushs_flush_without_global_atomics"] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=flags | ((128 + 4) << 8)))
    if "pushs" in ops:
        res["pushs_no_taps_but_an_atomic_for_every_slot"] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=(2 + 16 + 128) << 8))
    print("sigma", sigma, json.dumps({k: round(v, 3) for k, v in res.items()}))
