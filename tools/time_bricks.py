"""Kernel-level timing of the brick push on config 4 (8 sources 1x128^3 -> shared 512^3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol.distributed import push_count_shared
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
nsrc, n, m = 8, 128, 512
x = torch.randn(nsrc, 1, n, n, n, generator=g, device=dev)
gr = torch.randn([nsrc, n, n, n, 3], generator=g, device=dev).mul_(2.0)
gr += interpol.identity_grid([n, n, n], device=dev) * ((m - 1) / (n - 1))
for _ in range(3):
    push_count_shared(x, gr, [m, m, m], interpolation=3, bound="replicate", extrapolate=True, reduce="none")
torch.cuda.synchronize()
