"""push + count of the same grid: two launches vs INTERPOL_FLAG_WITH_COUNT (one pass), 4 x C x 256^3 cubic."""
import sys, os, json
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
res = {}
for sigma in (2.0, 0.0):
    for C in (1, 2, 3):
        inp, grid = bench.make_inputs(4, C, 256, sigma, dev, 1234)
        sep = timeit(lambda: (_hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1), _hip.scatter("count", None, grid, None, [3] * 3, [3] * 3, 1)))
        one = timeit(lambda: _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, with_count=True))
        res["sigma%g_C%d" % (sigma, C)] = {"push_then_count_ms": round(sep, 3), "with_count_ms": round(one, 3)}
        del inp, grid
print(json.dumps(res, indent=1))
