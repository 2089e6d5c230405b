import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch
from interpol import _hip
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(2)
for shape in ((64, 64, 64),):
    for b in ([0] * 3, [3] * 3, [6] * 3):
        for C in (1, 2):
            src = torch.randn(1, C, *shape, generator=g, device=dev)
            lin = [torch.linspace(2, n - 3, n, device=dev) for n in shape]
            grid = (torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None] + 0.5 * torch.randn(1, *shape, 3, generator=g, device=dev)).contiguous()
            for wc in (False, True):
                r = _hip.scatter("push", src, grid, list(shape), b, [1] * 3, 1, flags=_hip.FLAG_NO_FASTPATH, with_count=wc)
                a = _hip.scatter("push", src, grid, list(shape), b, [1] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER, with_count=wc)
                d = (a - r).abs()
                i = int(d.argmax())
                idx = torch.unravel_index(torch.tensor(i), d.shape)
                nb = int((d > 1e-4 * r.abs().max()).sum())
                print(b, C, wc, "max err", float(d.max()), "at", [int(v) for v in idx], "a", float(a.flatten()[i]), "r", float(r.flatten()[i]), "nbad", nb, "of", d.numel(), flush=True)
            rc = _hip.scatter("count", None, grid, list(shape), b, [1] * 3, 1, flags=_hip.FLAG_NO_FASTPATH)
            c = _hip.scatter("count", None, grid, list(shape), b, [1] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER)
            d = (c - rc).abs(); i = int(d.argmax())
            print(b, "count max err", float(d.max()), "c", float(c.flatten()[i]), "r", float(rc.flatten()[i]), "nbad", int((d > 1e-4).sum()), flush=True)
