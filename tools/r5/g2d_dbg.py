import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 1, 1, 64
for order in ([1, 1], [2, 2], [3, 3]):
    img = torch.randn(B, C, n, n, generator=g, device=dev)
    gout = torch.ones(B, C, n, n, device=dev)
    ident = interpol.identity_grid([n, n], device=dev)[None]
    grid = (ident + 0.3 * torch.randn(B, n, n, 2, generator=g, device=dev)).contiguous()
    backend.rough_deformations = True
    _, gg = _hip.pull_backward(gout, img, grid, [1, 1], order, 1, False, True)
    backend.rough_deformations = None
    _, rg = _hip.pull_backward(gout, img, grid, [1, 1], order, 1, False, True, flags=_hip.FLAG_NO_FASTPATH)
    print(order, gg[0, 10, 10:14].tolist(), rg[0, 10, 10:14].tolist())
    print("ratio", (gg / rg)[0, 20, 20:24].tolist())
