#!/usr/bin/env python
"""The windowed column prefilter (out of place) against the in-place kernels: parity over shapes / orders / bounds / dtypes, config 5's timing."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for shape in [(3, 200, 130), (2, 1024, 256), (1, 65, 64), (2, 70, 1000), (1, 2, 40, 300)]:
    for order in (2, 3):
        for bound in (0, 1, 2, 3):          # zero, replicate, dct1, dct2
            for dt in (torch.float32, torch.bfloat16):
                x = torch.randn(*shape, generator=g, device=dev).to(dt)
                dim = -2
                out = _hip.spline_filter_(torch.empty_like(x), bound, order, dim, src=x)          # windowed (out of place)
                ref = _hip.spline_filter_(x.double().clone(), bound, order, dim)                  # float64, in place
                err = float((out.double() - ref).abs().max() / ref.abs().max())
                tol = 2e-6 if dt == torch.float32 else 6e-3
                if not err < tol:
                    bad += 1
                    print("BAD", shape, order, bound, dt, err, flush=True)
print("parity: bad =", bad, flush=True)
B, C, n = 32, 3, 1024
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(B, C, n, n, generator=g, device=dev).to(dt)
    y = torch.empty_like(x)
    res = {"dtype": str(dt)}
    res["cols_out_of_place_ms"] = round(timeit(lambda: _hip.spline_filter_(y, 2, 2, -2, src=x)), 4)
    res["cols_in_place_ms"] = round(timeit(lambda: _hip.spline_filter_(y, 2, 2, -2)), 4)
    res["rows_in_place_ms"] = round(timeit(lambda: _hip.spline_filter_(y, 3, 3, -1)), 4)
    res["spline_coeff_nd_ms"] = round(timeit(lambda: interpol.spline_coeff_nd(x, [2, 3], ["dct1", "dct2"], 2)), 4)
    print(json.dumps(res), flush=True)
