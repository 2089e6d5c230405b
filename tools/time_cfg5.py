"""Config 5 (2-D, bf16 / fp32, orders [2,3]) pull and push: generic vs tiled kernels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
g = torch.Generator(device=dev).manual_seed(3)
B, C, n = 32, 3, 1024
for sigma in (2.0, 0.0):
    gr = torch.randn([B, n, n, 2], generator=g, device=dev).mul_(sigma) + interpol.identity_grid([n, n], device=dev)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(B, C, n, n, generator=g, device=dev).to(dt)
        for orders in ([2, 3], [3, 3], [1, 1]):
            for name, fl in (("generic", _hip.FLAG_NO_FASTPATH), ("default", 0), ("tiled", _hip.FLAG_FORCE_TILED)):
                tp = timeit(lambda: _hip.gather("pull", x, gr, [2, 5], orders, 1, flags=fl))
                ts = timeit(lambda: _hip.scatter("push", x, gr, None, [2, 5], orders, 1, flags=fl)) if name != "generic" else float("nan")
                print("sigma", sigma, str(dt)[6:], orders, name, "pull", round(tp, 3), "push", round(ts, 3))
