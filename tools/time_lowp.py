"""bf16 / f16 / f32 storage at 4 x 2 x 256^3 cubic dct2: pull, grad, push, backward."""
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
    inp32, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        inp = inp32.to(dt)
        r = {}
        r["pull"] = timeit(lambda: _hip.gather("pull", inp, grid, b3, o3, 1))
        r["grad"] = timeit(lambda: _hip.gather("grad", inp, grid, b3, o3, 1))
        r["push"] = timeit(lambda: _hip.scatter("push", inp, grid, None, b3, o3, 1))
        r["pull_bwd(both)"] = timeit(lambda: _hip.pull_backward(inp, inp, grid, b3, o3, 1, True, True))
        r["push_bwd(both)"] = timeit(lambda: _hip.push_backward(inp, inp, grid, b3, o3, 1, True, True))
        print("sigma", sigma, str(dt)[6:], {k: round(v, 2) for k, v in r.items()})
