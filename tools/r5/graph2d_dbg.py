import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip, backend
DEV = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(78)
n = 192
dt = torch.bfloat16 if len(sys.argv) < 2 else torch.float32
src = torch.randn([2, 3, n, n], generator=gen).to(dt).to(DEV)
ident = interpol.identity_grid([n, n])[None]
grid = (ident + 0.3 * torch.randn([2, n, n, 2], generator=gen)).to(DEV)
b, o = [2, 5], [2, 3]
fields = [ident + 0.3 * torch.randn([2, n, n, 2], generator=gen), ident + 9.0 * torch.randn([2, n, n, 2], generator=gen),
          (ident - n / 2) * 2.4 + n / 2 + 0.2 * torch.randn([2, n, n, 2], generator=gen)]
def check(tag, out, grid):
    ref = _hip.scatter("push", src.float(), grid, [n, n], b, o, 1, flags=_hip.FLAG_NO_FASTPATH)
    d = (out.float() - ref).abs()
    bad = (d > 0.05 * ref.abs().max()).nonzero()
    print(tag, "max err", float(d.max()), "bad", bad.shape[0], bad[:6].tolist(), flush=True)
for it, f in enumerate(fields):
    grid.copy_(f)
    check("eager %d" % it, _hip.scatter("push", src, grid, [n, n], b, o, 1), grid)
img = torch.randn([2, 3, n, n], generator=gen).to(dt).to(DEV)
mode = os.environ.get("RD", "auto")
backend.rough_deformations = {"auto": None, "tiles": False, "bricks": True}[mode]
print("mode", mode)
for rep in range(int(os.environ.get('SEQ', '0'))):
    for it, f in enumerate(fields):
        grid.copy_(f)
        x1 = _hip.gather("pull", img, grid, b, o, 1)
        out_push = _hip.scatter("push", src, grid, [n, n], b, o, 1)
        x2 = _hip.gather("pull", img, grid, b, o, 1)
        torch.cuda.synchronize()
        check("eager seq %d" % it, out_push, grid)
NF = _hip.FLAG_NO_FASTPATH
BR = _hip.FLAG_BINNED_SCATTER
variants = {"A": (NF, BR, NF), "B": (BR, NF, BR), "C": (BR, 0, BR), "D": (BR, BR, BR), "E": (0, BR, 0), "F": (NF, NF, NF)}
backend.rough_deformations = False      # flags decide
for name in os.environ.get("VARS", "ABCDEF"):
    f1, f2, f3 = variants[name]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _hip.gather("pull", img, grid, b, o, 1, flags=f1); _hip.scatter("push", src, grid, [n, n], b, o, 1, flags=f2); _hip.gather("pull", img, grid, b, o, 1, flags=f3)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x1 = _hip.gather("pull", img, grid, b, o, 1, flags=f1)
        out_push = _hip.scatter("push", src, grid, [n, n], b, o, 1, flags=f2)
        snap = out_push.clone()
        x2 = _hip.gather("pull", img, grid, b, o, 1, flags=f3)
        snap1 = x1.clone()
    for it, f in enumerate(fields):
        grid.copy_(f)
        g.replay(); torch.cuda.synchronize()
        check("graph %s %d" % (name, it), out_push, grid)
        check("   snap after op2  ", snap, grid)
        rp = _hip.gather("pull", img.float(), grid, b, o, 1, flags=NF)
        print("   pull errs x1 %.3g x2 %.3g" % (float((x1.float() - rp).abs().max()), float((x2.float() - rp).abs().max())), flush=True)
