#!/usr/bin/env python
"""What the call-level probe (csrc/push_owner.hip: own_probe) measures on the deformations of tools/rough_rows.py, next to the pull's
times: routed default / sample tiles with the per-tile hand-over only (debug bit 16384) / bricks alone.  The header of the workspace
is read back after a routed pull: gate, done, nslow (samples outside their tile's LDS box), nfar, nvalid, nbox, nfull, ncorner."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol, bench
from interpol import _hip, backend
dev = torch.device("cuda", 0)
def timeit(fn, reps=5, inner=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]
n = 256
inp, ident = bench.make_inputs(4, 2, n, 0.0, dev, 1234)
g = torch.Generator(device=dev).manual_seed(7)
cases = {"identity": ident}
for s in (1.0, 2.0, 3.0, 4.0, 6.0):
    cases["iid_sigma_%g" % s] = ident + s * torch.randn(ident.shape, generator=g, device=dev)
for amp in (2.0, 4.0, 8.0):
    ctrl = torch.randn(4, 3, 12, 12, 12, generator=g, device=dev) * amp
    disp = interpol.resize(ctrl, shape=[n] * 3, anchor="e", interpolation=3, bound="dct2", prefilter=True)
    cases["smooth_amp_%g" % amp] = ident + disp.permute(0, 2, 3, 4, 1)
    del ctrl, disp
for z in (1.5, 1.8, 2.0):
    cases["zoom_%g" % z] = (ident - (n - 1) / 2) * z + (n - 1) / 2
cases["stride_2_inside"] = ident * 0.5 + 20.0
captured = {}
real = _hip._optional_workspace
def spy(nbytes, dev_):
    ws = real(nbytes, dev_)
    captured["ws"] = ws
    return ws
_hip._optional_workspace = spy
only = sys.argv[1:]
for name, grid in cases.items():
    if only and name not in only:
        continue
    grid = grid.contiguous()
    pf = lambda fl=0: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)
    res = {}
    a = pf(); torch.cuda.synchronize()
    hdr = captured["ws"][:32].view(torch.int32).tolist() if captured.get("ws") is not None else None
    if hdr:
        res["verdict"] = hdr[0]; res["outside_pct"] = round(100.0 * hdr[2] / max(hdr[4], 1), 2); res["far_pct"] = round(100.0 * hdr[3] / max(hdr[4], 1), 2)
        res["box_mean"] = round(hdr[5] / max(hdr[6], 1)); res["corner_mean"] = round(hdr[7] / max(hdr[6], 1))
    r = pf(_hip.FLAG_NO_FASTPATH)
    res["rel_err"] = "%.1e" % float((a - r).abs().max() / r.abs().max())
    del a, r
    res["pull_default"] = round(timeit(pf), 3)
    res["pull_per_tile_only"] = round(timeit(lambda: pf(_hip.FLAG_AUTO_SCATTER | (16384 << 8))), 3)
    backend.rough_deformations = False
    res["pull_unrouted"] = round(timeit(pf), 3)
    backend.rough_deformations = None
    res["pull_tiles"] = round(timeit(lambda: pf(_hip.FLAG_FORCE_TILED)), 3)
    res["pull_bricks"] = round(timeit(lambda: pf(_hip.FLAG_BINNED_SCATTER)), 3)
    print(name, json.dumps(res), flush=True)
