import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 32, 3, 1024
x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
ident = interpol.identity_grid([n, n], device=dev)[None]
bc, o = [2, 5], [2, 3]
grid = (ident + 2.0 * torch.randn(B, n, n, 2, generator=g, device=dev)).contiguous()
backend.rough_deformations = None if os.environ.get('G2D_AUTO') else True
for _ in range(6):
    _hip.gather("pull", x, grid, bc, o, 1)
    _hip.pull_backward(x, x, grid, bc, o, 1, False, True)
torch.cuda.synchronize()
