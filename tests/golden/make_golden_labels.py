#!/usr/bin/env python
"""Golden vectors for the label-map path of grid_pull, generated from the
*reference itself* (interpol/api.py:194-205, prefilter=False).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_labels.py

Output: golden_labels.npz + golden_labels.json.  Data-generating test tooling; data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference          # noqa: E402

ref = sys.modules.get("interpol_ref") or load_reference()
ARR, CASES = {}, []


def main():
    g = torch.Generator().manual_seed(777)
    for dim, ishape, oshape in ((1, (9,), (23,)), (2, (7, 9), (11, 13)), (3, (5, 6, 7), (8, 7, 9))):
        for order in (0, 1, 2, 3):
            for bound in ("zero", "replicate", "dct2", "dft", "dst2"):
                for ex in (False, True):
                    if dim == 2 and order == 0 and not ex:
                        continue                    # reference bug B-1 (iso0.py:155, 2-D nearest mask): not reproduced
                    lab = torch.randint(-1, 4, [2, 2, *ishape], generator=g, dtype=torch.int64)
                    lin = [torch.linspace(-1.5, n + 0.5, m) for n, m in zip(ishape, oshape)]
                    grid = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)[None].repeat(2, *([1] * (dim + 1)))
                    grid = (grid + 0.8 * torch.randn(grid.shape, generator=g)).float()
                    # a few exact ties: half-way between voxels and exactly on voxels
                    flat = grid.reshape(2, -1, dim)
                    flat[0, 0] = 1.5
                    flat[0, 1] = 2.0
                    flat[1, 0] = 0.5
                    out = ref.grid_pull(lab, grid, interpolation=order, bound=bound, extrapolate=ex, prefilter=False)
                    i = len(CASES)
                    ARR["l%d/lab" % i] = lab.numpy().astype(np.int16)
                    ARR["l%d/grid" % i] = grid.numpy()
                    ARR["l%d/out" % i] = out.numpy().astype(np.int16)
                    CASES.append({"dim": dim, "order": order, "bound": bound, "extrapolate": ex, "i": i})
    np.savez_compressed(os.path.join(HERE, "golden_labels.npz"), **ARR)
    with open(os.path.join(HERE, "golden_labels.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_labels.py", "reference": "balbasty/torch-interpol @2024_10_08",
                   "cases": CASES}, f, indent=0)
    print(len(CASES), "cases,", sum(a.nbytes for a in ARR.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
