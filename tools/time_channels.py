"""pull / push at 4 x C x 256^3 cubic dct2 for C = 1..4 (per-channel cost of the tiled kernels)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for sigma in (2.0, 0.0):
    for C in (1, 2, 3, 4):
        inp, grid = bench.make_inputs(4, C, 256, sigma, dev, 1234)
        tp = timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1))
        tg = timeit(lambda: _hip.gather("grad", inp, grid, [3] * 3, [3] * 3, 1))
        ts = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1))
        print("sigma", sigma, "C", C, "pull", round(tp, 3), "grad", round(tg, 3), "push", round(ts, 3))
        del inp, grid
