import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
B, C, n = 32, 3, 1024
x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
xf = x.float()
for _ in range(6):
    interpol.spline_coeff_nd(x, [2, 3], ["dct1", "dct2"], 2)
    interpol.spline_coeff_nd(xf, [2, 3], ["dct1", "dct2"], 2)
torch.cuda.synchronize()
