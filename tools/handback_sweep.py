#!/usr/bin/env python
"""Hand-back threshold experiment: ms per call for threshold selectors 0..5 (dbg bits 9-11) and hand-back off (dbg 256)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench
dev = torch.device("cuda", 0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts)//2]
B, C, n = 4, 2, 256
g = torch.Generator(device=dev).manual_seed(11)
def scenarios():
    inp, grid0 = bench.make_inputs(B, C, n, 0.0, dev, 1234)
    for sigma in (3.0, 4.0, 6.0):
        yield "iid_sigma_%g" % sigma, inp, grid0 + sigma * torch.randn(grid0.shape, generator=g, device=dev)
    for amp in (4.0, 8.0):
        ctrl = torch.randn(B, 3, 12, 12, 12, generator=g, device=dev) * amp
        disp = interpol.resize(ctrl, shape=[n] * 3, anchor="e", interpolation=3, bound="dct2", prefilter=True)
        yield "smooth_amp_%g" % amp, inp, grid0 + disp.permute(0, 2, 3, 4, 1)
    for s in (1.75, 2.0, 3.0):   # zoom inside the field of view: a (256 s)^3 volume sampled with stride s
        m = int(n * s)
        big = torch.randn(1, C, m, m, m, generator=g, device=dev)
        yield "stride_%g_inside_fov" % s, big, (grid0[:1] * s).contiguous()
for name, inp, grid in scenarios():
    res = {}
    shape = list(inp.shape[2:])
    src = torch.randn([grid.shape[0], C] + list(grid.shape[1:4]), generator=g, device=dev)
    for label, fl in [("off", 256 << 8)] + [("thr%d" % (512 << sel), (sel << 9) << 8) for sel in range(3)] + [("generic", _hip.FLAG_NO_FASTPATH)]:
        res[label] = [round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=fl)), 2),
                      round(timeit(lambda: _hip.scatter("push", src, grid, shape, [3] * 3, [3] * 3, 1, flags=fl)), 2),
                      round(timeit(lambda: _hip.pull_backward(src, inp, grid, [3] * 3, [3] * 3, 1, False, True, flags=fl)), 2)]
    print("handback", name, "[pull, push, pullbwd_grid]", json.dumps(res), flush=True)
    del inp, grid, src
