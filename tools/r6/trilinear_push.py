import os, sys, json
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, bench
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=7, inner=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)/inner)
    ts.sort(); return round(ts[len(ts)//2],3)
for sigma in (2.0, 6.0):
    inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    print(json.dumps({"sigma": sigma, "trilinear_push": timeit(lambda: _hip.scatter("push", inp, grid, [256]*3, [3]*3, [1]*3, 1)),
                      "nearest_push": timeit(lambda: _hip.scatter("push", inp, grid, [256]*3, [3]*3, [0]*3, 1)),
                      "trilinear_count": timeit(lambda: _hip.scatter("count", None, grid, [256]*3, [3]*3, [1]*3, 1))}), flush=True)
