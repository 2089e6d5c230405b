#!/usr/bin/env python
"""A/B of two builds of the library inside one gpurun call at BASELINE config 3 (8 x 1 x 192^3 fp32, order 5, dft, sigma = 2): pull, grid_grad,
backward with both gradients, ms per call (default routing).  argv: <lib A> <lib B> [repeats]."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
    import torch, interpol
    from interpol import _hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(8, 1, 192, 192, 192, generator=g, device=dev)
    grid = torch.randn([8, 192, 192, 192, 3], generator=g, device=dev).mul_(2.0) + interpol.identity_grid([192] * 3, device=dev)
    src = torch.randn_like(x)
    def timeit(fn, reps=9, inner=3):
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
        return round(ts[len(ts) // 2], 4)
    b, o = [6] * 3, [5] * 3
    print(json.dumps({"pull": timeit(lambda: _hip.gather("pull", x, grid, b, o, 1)), "grad": timeit(lambda: _hip.gather("grad", x, grid, b, o, 1)),
                      "bwd_both": timeit(lambda: _hip.pull_backward(src, x, grid, b, o, 1, True, True)),
                      "push": timeit(lambda: _hip.scatter("push", src, grid, [192] * 3, b, o, 1)),
                      "push_bwd_both": timeit(lambda: _hip.push_backward(x, src, grid, b, o, 1, True, True))}))
    sys.exit(0)
libs, rep = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for _ in range(rep):
    for l in libs:
        env = dict(os.environ, INTERPOL_HIP_LIB=os.path.join(ROOT, "torch-interpol_amd", "lib", l))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True).stdout
        res[l].append(json.loads(out.strip().splitlines()[-1]))
print(json.dumps(res))
