"""Every operator of the path at 4 x 2 x 256^3 cubic dct2 (sigma = 2 and identity): a scan for outliers."""
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
b3, o3 = [3] * 3, [3] * 3
for sigma in (2.0, 0.0):
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    gout = torch.randn_like(inp)
    r = {}
    r["pull"] = timeit(lambda: _hip.gather("pull", inp, grid, b3, o3, 1))
    r["grad"] = timeit(lambda: _hip.gather("grad", inp, grid, b3, o3, 1))
    r["push"] = timeit(lambda: _hip.scatter("push", inp, grid, None, b3, o3, 1))
    r["count"] = timeit(lambda: _hip.scatter("count", None, grid, None, b3, o3, 1))
    r["pull_bwd(vol)"] = timeit(lambda: _hip.pull_backward(gout, inp, grid, b3, o3, 1, True, False))
    r["pull_bwd(grid)"] = timeit(lambda: _hip.pull_backward(gout, inp, grid, b3, o3, 1, False, True))
    r["pull_bwd(both)"] = timeit(lambda: _hip.pull_backward(gout, inp, grid, b3, o3, 1, True, True))
    r["push_bwd(val)"] = timeit(lambda: _hip.push_backward(gout, inp, grid, b3, o3, 1, True, False))
    r["push_bwd(both)"] = timeit(lambda: _hip.push_backward(gout, inp, grid, b3, o3, 1, True, True))
    r["count_bwd"] = timeit(lambda: _hip.push_backward(gout[:, :1], None, grid, b3, o3, 1, False, True))
    r["pull linear"] = timeit(lambda: _hip.gather("pull", inp, grid, b3, [1] * 3, 1))
    r["push linear"] = timeit(lambda: _hip.scatter("push", inp, grid, None, b3, [1] * 3, 1))
    r["pull nearest"] = timeit(lambda: _hip.gather("pull", inp, grid, b3, [0] * 3, 1))
    r["push nearest"] = timeit(lambda: _hip.scatter("push", inp, grid, None, b3, [0] * 3, 1))
    r["prefilter cubic 3 dims"] = timeit(lambda: interpol.spline_coeff_nd(inp, interpolation=3, bound="dct2", dim=3))
    print("sigma", sigma, {k: round(v, 2) for k, v in r.items()})
