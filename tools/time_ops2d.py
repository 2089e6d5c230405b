"""Every operator at config 5's shape (32 x 3 x 1024^2, orders [2,3], bf16 and fp32): outlier scan."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
g = torch.Generator(device=dev).manual_seed(3)
B, C, n = 32, 3, 1024
bb, oo = [2, 5], [2, 3]
grid = torch.randn([B, n, n, 2], generator=g, device=dev).mul_(2.0) + interpol.identity_grid([n, n], device=dev)
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(B, C, n, n, generator=g, device=dev).to(dt)
    r = {}
    r["pull"] = timeit(lambda: _hip.gather("pull", x, grid, bb, oo, 1))
    r["grad"] = timeit(lambda: _hip.gather("grad", x, grid, bb, oo, 1))
    r["push"] = timeit(lambda: _hip.scatter("push", x, grid, None, bb, oo, 1))
    r["count"] = timeit(lambda: _hip.scatter("count", None, grid, None, bb, oo, 1))
    r["pull_bwd(vol)"] = timeit(lambda: _hip.pull_backward(x, x, grid, bb, oo, 1, True, False))
    r["pull_bwd(grid)"] = timeit(lambda: _hip.pull_backward(x, x, grid, bb, oo, 1, False, True))
    r["pull_bwd(both)"] = timeit(lambda: _hip.pull_backward(x, x, grid, bb, oo, 1, True, True))
    r["push_bwd(val)"] = timeit(lambda: _hip.push_backward(x, x, grid, bb, oo, 1, True, False))
    r["push_bwd(both)"] = timeit(lambda: _hip.push_backward(x, x, grid, bb, oo, 1, True, True))
    r["pull cubic"] = timeit(lambda: _hip.gather("pull", x, grid, bb, [3, 3], 1))
    r["pull linear"] = timeit(lambda: _hip.gather("pull", x, grid, bb, [1, 1], 1))
    print(str(dt)[6:], {k: round(v, 2) for k, v in r.items()})
    for name, fl in (("generic", _hip.FLAG_NO_FASTPATH), ("default", 0)):
        print(name, "pull_bwd(grid)", round(timeit(lambda: _hip.pull_backward(x, x, grid, bb, oo, 1, False, True, flags=fl)), 2),
              "pull_bwd(both)", round(timeit(lambda: _hip.pull_backward(x, x, grid, bb, oo, 1, True, True, flags=fl)), 2),
              "push_bwd(both)", round(timeit(lambda: _hip.push_backward(x, x, grid, bb, oo, 1, True, True, flags=fl)), 2),
              "count_bwd", round(timeit(lambda: _hip.push_backward(x[:, :1], None, grid, bb, oo, 1, False, True, flags=fl)), 2))
