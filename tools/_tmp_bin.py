import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=4):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / inner)
    ts.sort(); return ts[len(ts) // 2]
for s in (2.0, 0.0, 6.0):
    inp, grid = bench.make_inputs(4, 2, 256, s, dev, 1234)
    gout = torch.randn_like(inp)
    f = lambda fl=0: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=fl)
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    gf = lambda fl=0: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, False, True, flags=fl)[1]
    B = _hip.FLAG_BINNED_SCATTER
    print("sigma", s, "push_owner", round(timeit(lambda: f(B)), 3), "push_default", round(timeit(f), 3), "pull_bricks", round(timeit(lambda: pf(B)), 3),
          "pull_default", round(timeit(pf), 3), "gradgrid_bricks", round(timeit(lambda: gf(B)), 3), "gradgrid_default", round(timeit(gf), 3), flush=True)
