#!/usr/bin/env python
"""Cubic label maps (row f4): the pass-per-distinct-label walk against the one-pass per-thread hash table (csrc/labels.hip)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=3):
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
g = torch.Generator(device=dev).manual_seed(11)
n = 192
ident = interpol.identity_grid([n, n, n], device=dev)[None]
gr = (ident + 2.0 * torch.randn(1, n, n, n, 3, generator=g, device=dev)).contiguous()
noise = torch.randint(0, 50, [1, 1, n, n, n], generator=g, device=dev, dtype=torch.int32)
many = torch.randint(-2**31, 2**31 - 1, [1, 1, n, n, n], generator=g, device=dev, dtype=torch.int64).to(torch.int32)
coarse = torch.randint(0, 50, [1, 1, n // 8, n // 8, n // 8], generator=g, device=dev, dtype=torch.int32)
blocky = coarse.repeat_interleave(8, 2).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
two = (torch.rand(1, 1, n, n, n, generator=g, device=dev) < 0.5).to(torch.int32)
for name, lab in (("iid 50 labels", noise), ("iid int32 labels", many), ("8^3 blocks", blocky), ("iid 2 labels", two)):
    res = {"labels": name}
    for bound, ex in (([3] * 3, 1), ([1, 4, 6], 0)):
        a = _hip.pull_labels(lab, gr, bound, [3] * 3, ex)
        h = _hip.pull_labels(lab, gr, bound, [3] * 3, ex, flags=2 << 8)
        res["equal_%d" % ex] = bool(torch.equal(a, h))
    res["walk_ms"] = round(timeit(lambda: _hip.pull_labels(lab, gr, [3] * 3, [3] * 3, 1, flags=2 << 8)), 3)
    res["hash_ms"] = round(timeit(lambda: _hip.pull_labels(lab, gr, [3] * 3, [3] * 3, 1)), 3)
    print(json.dumps(res), flush=True)
