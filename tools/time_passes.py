"""Per-pass timings of interpol_resample_1d (forward and adjoint) on the GPU box."""
import sys, os
sys.path.insert(0, "torch-interpol_amd"); sys.path.insert(0, ".")
import torch, interpol
from interpol import _hip
dev = "cuda"
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
y = torch.randn(4, 2, 256, 256, 256, device=dev)
lin = torch.arange(0., 256, device=dev) * 0.5 + 0.5 * (0.5 - 1)
for order in (1, 3):
    x = y
    for d in (-3, -2, -1):
        t = timeit(lambda: _hip.resample1d(x, lin, d, order, 3, 1, 0, adjoint=True, n_lattice=128))
        print("adjoint order", order, "dim", d, "in", tuple(x.shape), "ms", round(t, 3))
        x = _hip.resample1d(x, lin, d, order, 3, 1, 0, adjoint=True, n_lattice=128)
x = torch.randn(4, 2, 128, 128, 128, device=dev)
lin2 = torch.arange(0., 256, device=dev) * 0.5 + 0.5 * (0.5 - 1)
for d in (-1, -2, -3):
    t = timeit(lambda: _hip.resample1d(x, lin2, d, 3, 3, 1, 0))
    print("forward order 3 dim", d, "in", tuple(x.shape), "ms", round(t, 3))
    x = _hip.resample1d(x, lin2, d, 3, 3, 1, 0)
