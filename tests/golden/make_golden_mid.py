#!/usr/bin/env python
"""Mid-size golden vectors, generated from the *reference itself* (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mid.py

The vectors of make_golden.py hold < 4096 samples per case and therefore never reach the LDS-tile
kernels (the library routes tiny problems to the generic kernels).  These do: >= 4096 sample
points each, i.i.d. sigma = 2 voxel deformations (SURVEY 8d generator), so that the class-sorted
tiles (ops_sorted.hip) and the natural-order tiles (ops_tiled.hip) meet reference-generated data
directly -- the BASELINE.json configurations in miniature (18 - 32 per dim) incl. autograd, and a
bound / order / extrapolation sweep at 16 x 17 x 18.

Inputs are float32 (stored exactly); expected outputs are the reference's float64 results on the
same values, stored as float32 (relative storage error 6e-8, against a parity tolerance of 1e-5).
Outputs: golden_mid.npz + golden_mid.json.  Data-generating test tooling; the fixtures are data.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_DIR = "/root/reference/interpol"


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "interpol_ref", os.path.join(REFERENCE_DIR, "__init__.py"), submodule_search_locations=[REFERENCE_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["interpol_ref"] = mod
    spec.loader.exec_module(mod)
    return mod


ref = load_reference()
from interpol_ref import pushpull as P      # noqa: E402

ARR, CASES, BACKWARD = {}, [], []


def put(name, t, inp=False):
    a = np.ascontiguousarray(t.detach().cpu().numpy())
    if inp:
        assert np.array_equal(a.astype(np.float32).astype(np.float64), a), name
    ARR[name] = a.astype(np.float32)
    return name


def inputs(tag, B, C, sp, gsp, sigma, seed, bf16=False):
    """inp = randn, grid = identity scaled onto the volume + sigma * randn (voxels)."""
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn([B, C, *sp], generator=g, dtype=torch.float64).float()
    if bf16:
        inp = inp.bfloat16().float()                                  # bf16-representable values
    ident = ref.identity_grid(gsp, dtype=torch.float64)
    scale = torch.tensor([(n - 1) / max(m - 1, 1) for n, m in zip(sp, gsp)], dtype=torch.float64)
    grid = (ident[None] * scale + sigma * torch.randn([B, *gsp, len(sp)], generator=g, dtype=torch.float64)).float()
    return inp.double(), grid.double(), put("in/%s/inp" % tag, inp.double(), True), put("in/%s/grid" % tag, grid.double(), True)


def add(tag, op, inp, grid, nin, ngr, bound, order, ex, shape, storage="f32"):
    if op == "pull":
        out = P.grid_pull(inp, grid, bound, order, ex)
    elif op == "grad":
        out = P.grid_grad(inp, grid, bound, order, ex)
    elif op == "push":
        out = P.grid_push(inp, grid, shape, bound, order, ex)
    elif op == "count":
        out = P.grid_count(grid, shape, bound, order, ex)
    ins = {"grid": ngr} if op == "count" else {"inp": nin, "grid": ngr}
    key = "%s/%s/o%s/b%s/e%d" % (tag, op, "".join(map(str, order)), "".join(map(str, bound)), ex)
    CASES.append(dict(op=op, dim=grid.shape[-1], bound=list(bound), order=list(order), extrapolate=ex, shape=list(shape),
                      inputs=ins, output=put("out/" + key, out), storage=storage, tag=tag))


def backward(tag, fn, inp0, grid0, order, bound, shape=None, seed=0):
    """API-level autograd through the reference (api.py -> autograd.py:157-277)."""
    g = torch.Generator().manual_seed(seed)
    inp = inp0.clone().requires_grad_(True)
    grid = grid0.clone().requires_grad_(True)
    kw = dict(interpolation=order, bound=bound, extrapolate=True)
    y = ref.grid_pull(inp, grid, **kw) if fn == "grid_pull" else ref.grid_push(inp, grid, shape, **kw)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64).float().double()
    y.backward(gy)
    BACKWARD.append(dict(fn=fn, dim=grid0.shape[-1], interpolation=order, bound=bound, extrapolate=True, shape=shape,
                         inp=put("in/%s/%s_inp" % (tag, fn), inp0, True), grid=put("in/%s/%s_grid" % (tag, fn), grid0, True),
                         gout=put("in/%s/%s_gout" % (tag, fn), gy, True), out=put("out/%s/%s" % (tag, fn), y),
                         grad_inp=put("out/%s/%s_ginp" % (tag, fn), inp.grad), grad_grid=put("out/%s/%s_ggrid" % (tag, fn), grid.grad)))


def main():
    # ---- BASELINE configurations in miniature
    # cfg2: cubic / dct2, two channels, 20^3
    inp, grid, nin, ngr = inputs("cfg2", 1, 2, (20, 20, 20), (20, 20, 20), 2.0, 1234)
    for op in ("pull", "push", "count", "grad"):
        add("cfg2", op, inp, grid, nin, ngr, [3], [3], 1, (20, 20, 20))
    backward("cfg2", "grid_pull", inp, grid, 3, "dct2", seed=1)
    backward("cfg2", "grid_push", inp, grid, 3, "dct2", shape=[20, 20, 20], seed=2)
    # cfg3: order 5 / dft, one channel, 18^3: grad, pull and the backward of pull
    inp, grid, nin, ngr = inputs("cfg3", 1, 1, (18, 18, 18), (18, 18, 18), 2.0, 1235)
    for op in ("pull", "grad", "push"):
        add("cfg3", op, inp, grid, nin, ngr, [6], [5], 1, (18, 18, 18))
    backward("cfg3", "grid_pull", inp, grid, 5, "dft", seed=3)
    # cfg4: expanding push + count, 3 sources 1 x 12^3 -> 32^3, cubic / replicate
    inp, grid, nin, ngr = inputs("cfg4", 3, 1, (32, 32, 32), (12, 12, 12), 2.0, 1236)
    del ARR["in/cfg4/inp"]                                            # (only its shape placed the grid)
    g4 = torch.Generator().manual_seed(77)
    src = torch.randn([3, 1, 12, 12, 12], generator=g4, dtype=torch.float64).float().double()
    nsrc = put("in/cfg4/src", src, True)
    for op in ("push", "count"):
        add("cfg4", op, src, grid, nsrc, ngr, [1], [3], 1, (32, 32, 32))
    # cfg5: 2-D, mixed orders [2,3,5] -> [2,3] and bounds [dct1,dst2,zero] -> [dct1,dst2] (the reference truncates
    # the lists, jit_utils.py:9-15), three channels, 96 x 80; bf16-representable image values
    inp, grid, nin, ngr = inputs("cfg5", 1, 3, (80, 64), (80, 64), 2.0, 1237, bf16=True)
    for op in ("pull", "push"):
        add("cfg5", op, inp, grid, nin, ngr, [2, 5, 0], [2, 3, 5], 1, (80, 64), storage="bf16")
    # ---- sweep at 16 x 17 x 18 (4896 samples): orders 1..3 x all bounds, extrapolation modes, a single channel
    inp, grid, nin, ngr = inputs("sweep", 1, 1, (16, 17, 18), (16, 17, 18), 2.0, 1238)
    for order in (1, 2, 3):
        for bound in range(7):
            ex = (1, 0, 2)[(order + bound) % 3]
            for op in ("pull", "push", "count"):
                add("sweep", op, inp, grid, nin, ngr, [bound], [order], ex, (16, 17, 18))
    # a resampling (grid shape != volume shape), two channels, sigma 3
    inp, grid, nin, ngr = inputs("resamp", 1, 2, (17, 21, 13), (20, 18, 22), 3.0, 1239)
    for order, bound in ((3, 3), (2, 5), (3, 6), (3, 4)):
        add("resamp", "pull", inp, grid, nin, ngr, [bound], [order], 1, (17, 21, 13))
    add("resamp", "grad", inp, grid, nin, ngr, [4], [3], 0, (17, 21, 13))
    g5 = torch.Generator().manual_seed(78)
    src = torch.randn([1, 2, 20, 18, 22], generator=g5, dtype=torch.float64).float().double()
    nsrc = put("in/resamp/src", src, True)
    for order, bound in ((3, 3), (2, 5), (3, 6), (3, 4)):
        add("resamp", "push", src, grid, nsrc, ngr, [bound], [order], 1, (17, 21, 13))
    # smooth deformation (small box: one staging pass): sigma 0.3
    inp, grid, nin, ngr = inputs("smooth", 1, 2, (20, 18, 22), (20, 18, 22), 0.3, 1240)
    for op in ("pull", "push", "count"):
        add("smooth", op, inp, grid, nin, ngr, [3], [3], 1, (20, 18, 22))

    np.savez_compressed(os.path.join(HERE, "golden_mid.npz"), **ARR)
    with open(os.path.join(HERE, "golden_mid.json"), "w") as f:
        json.dump(dict(cases=CASES, backward=BACKWARD,
                       note="generated by tests/golden/make_golden_mid.py from balbasty/torch-interpol @2024_10_08"), f, indent=0)
    print(len(CASES), "operator cases,", len(BACKWARD), "backward cases,",
          os.path.getsize(os.path.join(HERE, "golden_mid.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
