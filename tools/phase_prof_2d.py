#!/usr/bin/env python
"""Per-phase cycle shares of the lean 2-D tile kernels at config 5 + kernel-only times (queue of 10 launches).
Phase shares need ops_tiled2d built with -DIP_PROF (INTERPOL_HIP_LIB=.../libinterpol_hip_prof.so)."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
NAMES = ["pull:build", "pull:stage", "pull:taps", "pull:store", "push:build", "push:density", "push:sources", "push:taps", "push:flush"]
B, C, n = 32, 3, 1024
gen = torch.Generator(device=dev).manual_seed(5)
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
x = torch.randn(B, C, n, n, generator=gen, device=dev).to(dt)
L = _hip.lib()
fn = getattr(L, "interpol_debug_prof_t2d_" + ("bf16" if dt == torch.bfloat16 else "f32"), None)
if fn is not None: fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for sigma in (2.0, 0.0):
    gr = torch.randn([B, n, n, 2], generator=gen, device=dev).mul_(sigma) + interpol.identity_grid([n, n], device=dev)
    ops = {"pull": lambda: _hip.gather("pull", x, gr, [2, 5], [2, 3], 1), "push": lambda: _hip.scatter("push", x, gr, None, [2, 5], [2, 3], 1)}
    for name, f in ops.items():
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        out = {"sigma": sigma, "op": name, "ms_per_call_queued": round(a.elapsed_time(b) / 10, 4)}
        if fn is not None:
            fn(None, 1); f(); torch.cuda.synchronize(); fn(buf, 1)
            tot = sum(buf)
            out["cycles_per_tile"] = round(tot / (B * 1024), 1)
            out["share"] = {NAMES[i]: round(buf[i] / tot, 4) for i in range(9) if buf[i]}
        print(json.dumps(out))
