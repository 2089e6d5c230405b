#!/usr/bin/env python
"""A/B of two builds of the library inside ONE gpurun call (the boxes differ by +-2 %): runs tools/r6/ab_flags.py in a subprocess per
library (INTERPOL_HIP_LIB), alternating, argv: <sigma> <lib A> <lib B> [repeats]."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sigma, libs, rep = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {l: [] for l in libs}
for _ in range(rep):
    for l in libs:
        env = dict(os.environ, INTERPOL_HIP_LIB=os.path.join(ROOT, "torch-interpol_amd", "lib", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r6", "ab_flags.py"), sigma, "0"], env=env, capture_output=True, text=True).stdout
        res[l].append(json.loads(out.strip().splitlines()[-1])["0"]["ms"])
print(json.dumps({l: {"ms": v, "median": sorted(v)[len(v) // 2]} for l, v in res.items()}))
