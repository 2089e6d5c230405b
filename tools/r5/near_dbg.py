import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch
from interpol import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(2)
shape = (40, 33, 50)
for b in ([0] * 3, [1] * 3, [2] * 3, [3] * 3, [4] * 3, [5] * 3, [6] * 3):
    for ex in (1, 2):
        src = torch.randn(1, 1, *shape, generator=g, device=dev)
        lin = [torch.linspace(-1, n, n, device=dev) for n in shape]
        grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + 0.3 * torch.randn(1, *shape, 3, generator=g, device=dev)).contiguous()
        r = _hip.scatter("push", src, grid, list(shape), b, [0] * 3, ex, flags=_hip.FLAG_NO_FASTPATH)
        a = _hip.scatter("push", src, grid, list(shape), b, [0] * 3, ex, flags=_hip.FLAG_BINNED_SCATTER)
        d = (a - r).abs()
        i = int(d.argmax())
        idx = [int(v) for v in torch.unravel_index(torch.tensor(i), d.shape)]
        print(b, ex, "max err", float(d.max()), "at", idx, "a", float(a.flatten()[i]), "r", float(r.flatten()[i]), "nbad", int((d > 1e-4).sum()), flush=True)
