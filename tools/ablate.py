#!/usr/bin/env python
"""Ablation timings of the tiled kernels on the GPU box (debug flags in interpol_problem.flags >> 8)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench

dev = torch.device("cuda", 0)
B, C, n = 4, 2, 256
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(B, C, n, sigma, dev, 1234)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)

res = {}
for name, flags in (("full", 0), ("no_stage_or_flush", 1 << 8), ("no_compute", 2 << 8), ("neither", 3 << 8), ("generic", 1)):
    if name == "generic" and "--generic" not in sys.argv:
        continue
    res["pull_" + name] = timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=flags))
    res["push_" + name] = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=flags))
res["memset_537MB"] = timeit(lambda: torch.empty(B, C, n, n, n, device=dev).zero_())
res["copy_537MB"] = timeit(lambda: inp.clone())
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
res2 = {"pull_pairmode": timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)),
        "pull_single_channel_mode": timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=4 << 8))}
print(json.dumps({k: round(v, 3) for k, v in res2.items()}))
