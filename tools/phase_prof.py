#!/usr/bin/env python
"""Per-phase cycle shares of the tiled push kernel (needs ops_tiled built with -DIP_PROF:
   rm torch-interpol_amd/build/ops_tiled_f32_[12].o && make -C torch-interpol_amd PROF=1)."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench

NAMES = ["build", "pair:slow", "pair:zero", "pair:taps", "pair:flush", "pull:stage", "pull:gather", "pull:store+slow", "single-channel path", "build:load+minmax", "build:tables+classify", "build:density", "build:issue coord loads", "build:wait coord loads", "build:channel maxima"]
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
L = _hip.lib()
FN = {"pull": L.interpol_debug_prof_f32_1, "push": L.interpol_debug_prof_f32_2}
for f in FN.values():
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
def run(op):
    if op == "push":
        _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1)
    else:
        _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1)
    torch.cuda.synchronize()
for op in ("push",):          # (the pull of config 2 runs in ops_sorted.hip: tools/phase_prof_sorted.py)
  fn = FN[op]
  run(op); fn(None, 1); run(op); fn(buf, 1)
  tot = sum(buf)
  print(op, json.dumps({"sigma": sigma, "total_cycles_per_block_sum": tot,
                  "share": {NAMES[i] if i < len(NAMES) else str(i): round(buf[i] / tot, 4) for i in range(16) if buf[i]}}))
