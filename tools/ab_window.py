#!/usr/bin/env python
"""A/B of the pull organisations: the default (four-pass class-sorted tiles + compact tiles chosen per tile), the same
without compact tiles (debug bit 8192), the windowed gather (csrc/pull_window.hip, debug bit 4096) -- against the generic
kernels on a spread of problems, then config-2 timings.  usage: tools/ab_window.py [quick]"""
import os, sys, json, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench
dev = torch.device("cuda", 0)
OLD = 8192 << 8          # no compact tiles
WIN = 4096 << 8          # the windowed gather
BOUNDS = {"zero": 0, "replicate": 1, "dct1": 2, "dct2": 3, "dst1": 4, "dst2": 5, "dft": 6}


def run_case(shape_in, shape_out, B, C, sigma, bound, order, extrap, dtype=torch.float32, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn([B, C] + list(shape_in), generator=g).to(dev).to(dtype)
    ident = torch.stack(torch.meshgrid(*[torch.arange(float(s)) for s in shape_out], indexing="ij"), -1)
    grid = (ident[None] * scale + shift + sigma * torch.randn([B] + list(shape_out) + [3], generator=g)).to(dev)
    new = _hip.gather("pull", inp, grid, bound, [order] * 3, extrap)
    old = _hip.gather("pull", inp, grid, bound, [order] * 3, extrap, flags=OLD)
    ref = _hip.gather("pull", inp, grid, bound, [order] * 3, extrap, flags=_hip.FLAG_NO_FASTPATH)
    sc = float(ref.float().abs().max()) + 1e-30
    e_new = float((new.float() - ref.float()).abs().max()) / sc
    e_old = float((old.float() - ref.float()).abs().max()) / sc
    return e_new, e_old


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    worst = 0.0
    nfail = 0
    cases = []
    for (n_in, n_out) in [((48, 48, 48), (48, 48, 48)), ((40, 33, 50), (37, 45, 29)), ((64, 64, 64), (32, 32, 32))]:
        for sigma in [0.0, 0.7, 2.0, 5.0]:
            for bound in ["dct2", "dft", "zero", "dst1", "replicate", "dct1", "dst2"]:
                for order in [3, 2]:
                    cases.append((n_in, n_out, sigma, bound, order))
    if quick:
        cases = cases[::7]
    k = 0
    for (n_in, n_out, sigma, bound, order) in cases:
        k += 1
        extrap = [1, 0, 2][k % 3]
        Cc = [2, 1, 3][k % 3]
        dt = [torch.float32, torch.float32, torch.bfloat16][k % 3] if k % 5 == 0 else torch.float32
        b = [BOUNDS[bound]] * 3
        try:
            e_new, e_old = run_case(n_in, n_out, 2, Cc, sigma, b, order, extrap, dt, seed=k,
                                    scale=(n_in[0] - 1) / max(n_out[0] - 1, 1) if k % 4 == 0 else 1.0, shift=-1.5 if k % 6 == 0 else 0.0)
        except Exception as ex:  # noqa
            print("EXC", n_in, n_out, sigma, bound, order, repr(ex)); nfail += 1; continue
        tol = 2e-2 if dt != torch.float32 else 2e-5
        bad = not (e_new <= tol)
        worst = max(worst, e_new if dt == torch.float32 else 0.0)
        if bad:
            nfail += 1
            print("FAIL", n_in, n_out, "sigma", sigma, bound, "order", order, "extrap", extrap, "C", Cc, dt, "new", e_new, "old", e_old)
    print("cases", len(cases), "failures", nfail, "worst fp32 rel err (new vs generic)", worst)

    def timeit(fn, reps=7, batch=4):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(batch): fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / batch)
        ts.sort(); return ts[len(ts) // 2]
    res = {}
    for sigma in [2.0, 0.0, 0.5, 1.0]:
        inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
        for _ in range(10): _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)
        res["sigma%g" % sigma] = {
            "default": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)), 3),
            "no_compact": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=OLD)), 3),
            "window": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=WIN)), 3),
            "default_quadratic": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [2] * 3, 1)), 3),
            "default_nostage": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=1 << 8)), 3),
            "default_notaps": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=2 << 8)), 3),
            "default_neither": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=3 << 8)), 3),
        }
        a = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)
        b_ = _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=OLD)
        res["sigma%g" % sigma]["maxdiff_new_old"] = float((a - b_).abs().max())
        del inp, grid, a, b_
    grid = bench.smooth_grid(4, 256, 2.0, dev, 7)
    inp = torch.randn([4, 2, 256, 256, 256], device=dev)
    res["smooth"] = {"default": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)), 3),
                     "no_compact": round(timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=OLD)), 3)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
