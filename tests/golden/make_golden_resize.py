#!/usr/bin/env python
"""Golden vectors for `resize` / `restrict`, generated from the *reference
itself* (interpol/resize.py, interpol/restrict.py).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_resize.py

Inputs are float32-representable.  Both functions build their sampling lattice in the
image's dtype, so every case stores TWO expectations: `out64` (reference on the float64
input) and `out32` (reference on the float32 input: float32 lattice).  Output:
golden_resize.npz + golden_resize.json.  Data-generating test tooling; fixtures are data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference          # noqa: E402  (also the loader of the other fixtures)

ref = sys.modules.get("interpol_ref") or load_reference()

ARR, CASES = {}, []


def put(name, t, dtype):
    ARR[name] = np.ascontiguousarray(t.detach().cpu().numpy()).astype(dtype)
    return name


def add(fn, x64, kw):
    i = len(CASES)
    x32 = x64.float()
    f = getattr(ref, fn)
    out64 = f(x64, **kw)
    out32 = f(x32, **kw)
    CASES.append({
        "fn": fn, "kwargs": kw, "inp": put("r%d/inp" % i, x32, np.float32),
        "out64": put("r%d/out64" % i, out64, np.float64), "out32": put("r%d/out32" % i, out32, np.float32),
        "shape": list(out64.shape)})


def main():
    g = torch.Generator().manual_seed(4242)
    x3 = torch.randn([2, 2, 5, 6, 7], generator=g, dtype=torch.float64).float().double()
    x2 = torch.randn([1, 2, 8, 9], generator=g, dtype=torch.float64).float().double()
    x1 = torch.randn([2, 1, 13], generator=g, dtype=torch.float64).float().double()
    for fn in ("resize", "restrict"):
        up = fn == "resize"
        for anchor in ("c", "e", "f", "l"):
            for order in (1, 3):
                for bound in ("dct2", "replicate", "dft"):
                    # (a scalar factor resizes the LAST dim only: nb_dim = len(make_list(factor)), resize.py:74)
                    kw = dict(factor=2, anchor=anchor, interpolation=order, bound=bound)
                    if up:      # the reference's prefilter exists for dct1 / dct2 / dft only (coeff.py:230-256)
                        kw["prefilter"] = order > 1 and bound != "replicate"
                    if bound == "dct2" or anchor == "c":
                        add(fn, x3, dict(kw, factor=[2, 2, 2]) if order == 3 else dict(kw))
                    add(fn, x2, dict(kw, factor=[2, 3] if up else [2, 3]))
        # explicit shapes, non-integer factors, mixed anchors and orders, 1-D, order 0 / 2 / 5
        add(fn, x3, dict(shape=[8, 5, 11] if up else [3, 4, 5], anchor="c", interpolation=2, bound="dct1"))
        add(fn, x3, dict(factor=[1.5, 2.0, 1.25], anchor=["e", "f", "c"] if up else "e", interpolation=[1, 2, 3], bound=["dct2", "dst2", "zero"],
                         **({"prefilter": False} if up else {})))
        add(fn, x2, dict(factor=1.7, anchor="e", interpolation=5, bound="dft"))
        add(fn, x2, dict(factor=2, anchor="f", interpolation=0, bound="replicate", **({"prefilter": False} if up else {})))
        add(fn, x1, dict(factor=3, anchor="c", interpolation=3, bound="dct2"))
        add(fn, x1, dict(shape=[7], anchor="e", interpolation=1, bound="zero", extrapolate=False, **({"prefilter": False} if up else {})))
    if True:
        add("restrict", x3, dict(factor=2, anchor="e", interpolation=1, bound="dct2", reduce_sum=True))
    np.savez_compressed(os.path.join(HERE, "golden_resize.npz"), **ARR)
    with open(os.path.join(HERE, "golden_resize.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_resize.py", "reference": "balbasty/torch-interpol @2024_10_08",
                   "cases": CASES}, f, indent=0)
    print(len(CASES), "cases,", sum(a.nbytes for a in ARR.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
