import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for (B, C, n0, n1) in [(2, 3, 200, 131), (1, 5, 97, 140)]:
    ident = interpol.identity_grid([n0, n1], device=dev)[None]
    for sigma in (1.0, 9.0):
        for bound, order, ex in (([2, 5], [2, 3], 1), ([0, 6], [3, 1], 0), ([1, 3], [1, 1], 2)):
            for dt in (torch.float32, torch.bfloat16):
                disp = (sigma * torch.randn(B, n0, n1, 2, generator=g, device=dev)).contiguous()
                grid = (ident + disp).contiguous()
                img = torch.randn(B, C, n0, n1, generator=g, device=dev).to(dt)
                tol = 4e-6 if dt == torch.float32 else 8e-3
                for rd in (True, None):
                    backend.rough_deformations = rd
                    D = _hip.FLAG_DISPLACEMENT
                    pairs = [(_hip.gather("pull", img, disp, bound, order, ex, flags=D), _hip.gather("pull", img, grid, bound, order, ex)),
                             (_hip.scatter("push", img, disp, [n0, n1], bound, order, ex, flags=D, with_count=True), _hip.scatter("push", img, grid, [n0, n1], bound, order, ex, with_count=True)),
                             (_hip.scatter("count", None, disp, [n0, n1], bound, order, ex, flags=D), _hip.scatter("count", None, grid, [n0, n1], bound, order, ex)),
                             (_hip.pull_backward(img, img, disp, bound, order, ex, False, True, flags=D)[1], _hip.pull_backward(img, img, grid, bound, order, ex, False, True)[1])]
                    for i, (a, r) in enumerate(pairs):
                        e = float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-30))
                        if not e < tol:
                            bad += 1; print("BAD", i, B, C, sigma, bound, order, ex, dt, rd, e, flush=True)
backend.rough_deformations = None
print("displacement: bad =", bad)
