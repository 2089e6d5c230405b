"""Per-dim timings of the spline prefilter on config 5 (32x3x1024^2)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
dev = torch.device("cuda", 0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(32, 3, 1024, 1024, device=dev).to(dt)
    for order in (2, 3, 5):
        for dim in (-1, -2):
            t = timeit(lambda: interpol.spline_coeff(x, interpolation=order, bound="dct2", dim=dim))
            tc = timeit(lambda: x.clone())
            print(dt, "order", order, "dim", dim, "ms", round(t, 3), "(clone alone", round(tc, 3), ")")
