#!/usr/bin/env python
"""Differential fuzz of separable lattices (interpol.SeparableGrid -> INTERPOL_FLAG_SEPARABLE_GRID: the D coordinate vectors travel
instead of a grid tensor) through the default routing vs the generic kernels on the dense grid they stand for: strides 0.5 - 3
(stretched tiles, hand-back), orders 1-3, every bound, extrapolation modes.  usage: tools/fuzz_separable.py"""
import sys, random
sys.path.insert(0, "torch-interpol_amd"); sys.path.insert(0, ".")
import torch, interpol
from interpol import _hip
from interpol.sepgrid import SeparableGrid
dev = torch.device("cuda", 0)
rnd = random.Random(4); gen = torch.Generator().manual_seed(4)
def rel(a, r): return float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-20))
bad = 0
for case in range(150):
    dim = rnd.choice([2, 3, 3]); B, C = rnd.choice([1, 2]), rnd.choice([1, 2, 3])
    ishape = [rnd.randint(20, 60) for _ in range(3)] if dim == 3 else [rnd.randint(40, 200) for _ in range(2)]
    oshape = [rnd.randint(17, 50) for _ in range(3)] if dim == 3 else [rnd.randint(65, 200) for _ in range(2)]
    order = [rnd.choice([1, 2, 3])] * dim; bound = [rnd.randrange(7)] * dim; ex = rnd.choice([0, 1, 1, 2])
    stride = rnd.choice([0.5, 1.0, 1.0, 2.2, 3.0])
    lins = [(torch.arange(n, dtype=torch.float32) * stride + rnd.uniform(-1, 1)).to(dev) for n in oshape]
    sg = SeparableGrid(lins); dense = sg.dense().contiguous()
    vol = torch.randn([B, C, *ishape], generator=gen).to(dev); src = torch.randn([B, C, *oshape], generator=gen).to(dev)
    for rep in range(2):
        checks = [("pull", rel(_hip.gather("pull", vol, sg, bound, order, ex), _hip.gather("pull", vol, dense, bound, order, ex, flags=_hip.FLAG_NO_FASTPATH))),
                  ("grad", rel(_hip.gather("grad", vol, sg, bound, order, ex), _hip.gather("grad", vol, dense, bound, order, ex, flags=_hip.FLAG_NO_FASTPATH))),
                  ("push", rel(_hip.scatter("push", src, sg, ishape, bound, order, ex), _hip.scatter("push", src, dense, ishape, bound, order, ex, flags=_hip.FLAG_NO_FASTPATH))),
                  ("count", rel(_hip.scatter("count", None, sg, ishape, bound, order, ex), _hip.scatter("count", None, dense, ishape, bound, order, ex, flags=_hip.FLAG_NO_FASTPATH)))]
    fails = [(k, "%.1e" % v) for k, v in checks if not v <= 3e-4]
    if fails: bad += 1; print("MISMATCH", case, dim, B, C, ishape, oshape, order, bound, ex, stride, fails)
print("separable lattices: 150 cases,", bad, "bad")
