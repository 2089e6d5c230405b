import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
def err(a, r):
    return float((a - r).abs().max() / r.abs().max())
shape = (96, 96, 96)
ident = interpol.identity_grid(shape)[None]
for C in (1, 2):
  for order in ([1]*3, [2]*3, [3]*3, [1,2,3], [3,1,2], [4]*3, [5]*3, [6]*3, [7]*3, [2,3,5]):
    for sigma in (4.0,):
        vol = torch.randn([2, C, *shape], generator=gen).to(dev)
        gout = torch.randn([2, C, *shape], generator=gen).to(dev)
        grid = (ident + sigma * torch.randn([2, *shape, 3], generator=gen)).contiguous().to(dev)
        b = [3] * 3
        ref = _hip.pull_backward(gout, vol, grid, b, order, 1, False, True, flags=_hip.FLAG_NO_FASTPATH)[1]
        out = []
        for fl in (0, _hip.FLAG_FORCE_TILED, _hip.FLAG_FORCE_TILED | (16 << 8)):
            got = _hip.pull_backward(gout, vol, grid, b, order, 1, False, True, flags=fl)[1]
            out.append("%.1e" % err(got, ref))
        print("C", C, "order", order, "sigma", sigma, "default / FORCE_TILED / FORCE_TILED+dbg16:", out, flush=True)
