"""Tiled vs generic kernels per spline order (2 x 2 x 160^3, dct2): is the fast path the faster one everywhere?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for sigma in (2.0, 0.0):
    inp, grid = bench.make_inputs(2, 2, 160, sigma, dev, 7)
    for K in range(1, 8):
        o = [K] * 3
        row = []
        for fl in (0, _hip.FLAG_NO_FASTPATH):
            row.append(round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, o, 1, flags=fl)), 2))
            row.append(round(timeit(lambda: _hip.gather("grad", inp, grid, [3] * 3, o, 1, flags=fl)), 2))
            if fl == 0 or K <= 3:
                row.append(round(timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, o, 1, flags=fl), 1 if fl else 3), 2))
            else:
                row.append(float("nan"))
        print("sigma", sigma, "K", K, "tiled pull/grad/push", row[:3], "generic", row[3:])
