import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
y = torch.randn(4, 2, 256, 256, 256, generator=g, device=dev)
x = torch.randn(4, 2, 128, 128, 128, generator=g, device=dev)
for _ in range(5):
    interpol.restrict(y, factor=[2, 2, 2], anchor='e', interpolation=1, bound='dct2')
    interpol.restrict(y, factor=[2, 2, 2], anchor='e', interpolation=3, bound='dct2')
    interpol.resize(x, factor=[2, 2, 2], anchor='e', interpolation=3, bound='dct2', prefilter=False)
torch.cuda.synchronize()
