import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
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
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 32, 3, 1024
x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
ident = interpol.identity_grid([n, n], device=dev)[None]
bc, o = [2, 5], [2, 3]
grid = (ident + 2.0 * torch.randn(B, n, n, 2, generator=g, device=dev)).contiguous()
for dbg in [0, 1, 2, 4, 3, 5, 7]:
    f = _hip.FLAG_BINNED_SCATTER | (dbg << 8)
    print(json.dumps({"dbg": dbg, "pull": round(timeit(lambda: _hip.gather("pull", x, grid, bc, o, 1, flags=f)), 3)}), flush=True)
