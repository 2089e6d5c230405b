import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
shape = (96, 96, 96)
ident = interpol.identity_grid(shape)[None]
order = [3] * 3
vol = torch.randn([2, 1, *shape], generator=gen).to(dev)
gout = torch.randn([2, 1, *shape], generator=gen).to(dev)
grid = (ident + 4.0 * torch.randn([2, *shape, 3], generator=gen)).contiguous().to(dev)
b = [3] * 3
ref = _hip.pull_backward(gout, vol, grid, b, order, 1, False, True, flags=_hip.FLAG_NO_FASTPATH)[1]
EXTRA = int(sys.argv[1]) if len(sys.argv) > 1 else 0
got = _hip.pull_backward(gout, vol, grid, b, order, 1, False, True, flags=_hip.FLAG_FORCE_TILED | ((16 | EXTRA) << 8))[1]
bad = ((got - ref).abs() > 1e-4 * ref.abs().max()).any(-1)          # (B, X, Y, Z)
print("bad samples", int(bad.sum()), "of", bad.numel())
# per tile of 16^3: fraction bad
t = bad.view(2, 6, 16, 6, 16, 6, 16).float().mean((2, 4, 6))
print("tiles with bad samples:", int((t > 0).sum()), "of", t.numel())
idx = (t > 0).nonzero()
print(idx[:40].tolist())
print("bad fraction in those tiles:", t[t > 0][:20].tolist())
# first-tap distance: are the bad samples the far ones?
disp = (grid - ident.to(dev)).abs().amax(-1)
print("mean |disp|max of bad samples %.2f, of good samples %.2f" % (float(disp[bad].mean()), float(disp[~bad].mean())))
z = (got == 0).all(-1)
print("bad samples that are exactly zero:", int((z & bad).sum()))
if bad.any():
    tb = (t > 0).nonzero()[1].tolist()          # second bad tile
    bb, tx, ty, tz = tb
    sub = bad[bb, 16 * tx:16 * tx + 16, 16 * ty:16 * ty + 16, 16 * tz:16 * tz + 16]
    print("tile", tb, "bad per local x:", sub.sum((1, 2)).tolist())
    print("bad per local y:", sub.sum((0, 2)).tolist())
    print("bad per local z:", sub.sum((0, 1)).tolist())
    # thread id of a sample: tid = (x % 2) * 256 + y * 16 + z ; v = x // 2   (XSTEP = 512 / 256 = 2)
    xs, ys, zs = sub.nonzero(as_tuple=True)
    tids = (xs % 2) * 256 + ys * 16 + zs
    waves = (tids // 64)
    print("bad per wave:", torch.bincount(waves, minlength=8).tolist())
    print("bad per v:", torch.bincount(xs // 2, minlength=8).tolist())
