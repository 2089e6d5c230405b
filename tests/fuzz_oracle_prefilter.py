#!/usr/bin/env python
"""Randomised parity run of the prefilter (spline_coeff_nd, csrc/prefilter.hip) against the oracle (hand-run on a GPU box;
not collected by pytest): 1-3 filtered dims under 0-2 leading dims, line lengths 1..700 (the register-line kernels want 64 R,
everything else takes the chunked / serial kernels), orders 0-7 per dim, the bounds the reference implements, f32 / f64 / bf16,
in place and out of place.  usage: python tests/fuzz_oracle_prefilter.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, interpol
from oracle import oracle
dev = torch.device("cuda", 0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); gen = torch.Generator().manual_seed(seed)
oracle.set_threads(min(16, os.cpu_count() or 1))
NAMES = {0: "zero", 1: "replicate", 2: "dct1", 3: "dct2", 6: "dft"}
bad = 0
for case in range(n_cases):
    nd = rnd.choice([1, 2, 2, 3])
    lens = [rnd.choice([1, 2, 3, 5, 17, 63, 64, 65, 128, 200, 256, 257, 512, 700]) if rnd.random() < 0.6 else rnd.randint(1, 300) for _ in range(nd)]
    while np.prod(lens) > 4e6: lens[rnd.randrange(nd)] = rnd.randint(1, 64)
    lead = [rnd.randint(1, 4) for _ in range(rnd.choice([0, 1, 2]))]
    order = [rnd.randint(0, 7) for _ in range(nd)]
    bound = [rnd.choice([0, 1, 2, 3, 6]) for _ in range(nd)]
    dt = rnd.choice([torch.float32, torch.float32, torch.float64, torch.bfloat16])
    inplace = rnd.random() < 0.3
    x = torch.randn(lead + lens, generator=gen).to(dt)
    want = oracle.spline_coeff_nd(x.double().numpy(), bound, order, dim=nd)
    try:
        xin = x.to(dev).clone()
        got = interpol.spline_coeff_nd(xin, order, [NAMES[b] for b in bound], nd, inplace=inplace)
        if inplace and got.data_ptr() != xin.data_ptr(): raise RuntimeError("inplace result is a different tensor")
        if not inplace and not torch.equal(xin.cpu(), x): raise RuntimeError("out-of-place call modified its input")
    except Exception as e:
        print("EXCEPTION", case, lead, lens, order, bound, dt, inplace, repr(e)); bad += 1; continue
    err = float(np.abs(got.double().cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-20))
    tol = {torch.float32: 2e-5, torch.float64: 1e-12, torch.bfloat16: 3e-2}[dt] * (1 + 2 * sum(o >= 2 for o in order))
    if not err <= tol:
        bad += 1; print("MISMATCH", case, "lead", lead, "lens", lens, "order", order, "bound", bound, dt, "inplace", inplace, "err %.2e" % err)
print("fuzz prefilter vs oracle: %d cases, %d bad (seed %d)" % (n_cases, bad, seed))
sys.exit(1 if bad else 0)
