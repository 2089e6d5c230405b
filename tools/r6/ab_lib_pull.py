#!/usr/bin/env python
"""A/B of two builds of the library inside one gpurun call: grid_pull and grid_push at config 2 (default routing), ms per call.
argv: <sigma> <lib A> <lib B> [repeats]."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
    import torch, interpol, bench
    dev = torch.device("cuda", 0)
    inp, grid = bench.make_inputs(4, 2, 256, float(sys.argv[2]), dev, 1234)
    def timeit(fn, reps=9, inner=4):
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
    kw = dict(interpolation=3, bound="dct2", extrapolate=True)
    print(json.dumps({"pull": round(timeit(lambda: interpol.grid_pull(inp, grid, **kw)), 4), "push": round(timeit(lambda: interpol.grid_push(inp, grid, **kw)), 4)}))
    sys.exit(0)
sigma, libs, rep = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {l: [] for l in libs}
for _ in range(rep):
    for l in libs:
        env = dict(os.environ, INTERPOL_HIP_LIB=os.path.join(ROOT, "torch-interpol_amd", "lib", l))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sigma], env=env, capture_output=True, text=True).stdout
        res[l].append(json.loads(out.strip().splitlines()[-1]))
print(json.dumps(res))
