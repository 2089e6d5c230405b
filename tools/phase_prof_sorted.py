#!/usr/bin/env python
"""Per-phase cycle shares of the class-sorted tile kernels.  Needs a library whose ops_sorted
objects were built with -DIP_PROF (INTERPOL_HIP_LIB=.../libinterpol_hip_prof.so)."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol import _hip
import bench

NAMES = {"pullw": ["setup:read records", "stage (all windows)", "taps (all windows)", "unsort+store+slow", "setup:coords+split+minmax", "setup:tables+classify",
                   "setup:planner", "setup:records to sorted places"],
         "pull": ["build:read sorted", "stage", "taps", "unsort+store+slow", "build:coords", "build:split+minmax", "build:tables+classify", "build:scan+holes", "build:records"],
         "push": ["bin:load", "bin:brick+rank", "bin:scan+desc", "bin:direct+pos", "bin:exchange", "bin:store", "6", "7",
                  "acc:desc", "acc:pass1", "acc:density", "acc:taps", "acc:flush", "acc:stencil counts", "acc:draw+ndesc", "acc:classes+queue"],
         "pushs": ["build+density", "taps (4 passes)", "flush (4 passes)", "slow + tail", "sources+scale", "build:coords", "build:split+minmax", "build:tables+classify", "build:scan+holes", "build:records"]}
dev = torch.device("cuda", 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
L = _hip.lib()
# pull: pull_sorted; push: the owner-computes push (FLAG_BINNED_SCATTER); pushs: push_sorted (dbg switch 128)
fn = L.interpol_debug_prof_owner if (sys.argv[2:] and sys.argv[2] == "push") else (L.interpol_debug_prof_window_f32 if (sys.argv[2:] and sys.argv[2] == "pullw") else L.interpol_debug_prof_sorted_f32)
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
def run(op):
    if op == "push":
        _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=_hip.FLAG_BINNED_SCATTER | ((int(sys.argv[3]) if len(sys.argv) > 3 else 0) << 8))
    elif op == "pushs":
        _hip.scatter("push", inp, grid, None, [3] * 3, [3] * 3, 1, flags=128 << 8)
    else:      # pullw: the windowed gather (the default); pull: the four-pass tiles (debug bit 4096)
        _hip.gather("pull", inp, grid, [3] * 3, [3] * 3, 1, flags=((int(sys.argv[3]) if len(sys.argv) > 3 else 0) | (4096 if op == "pull" else 0)) << 8)
    torch.cuda.synchronize()
for op in sys.argv[2:] or ["pull"]:
    flags = int(sys.argv[3]) << 8 if len(sys.argv) > 3 else 0
    run(op); fn(None, 1); run(op); fn(buf, 1)
    tot = sum(buf)
    names = NAMES[op]
    nblk = 16384 if op == "pullw" else 512           # (one workgroup per tile / persistent workgroups)
    print(op, json.dumps({"sigma": sigma, "cycles_per_block_avg": tot / nblk,
          "share": {names[i] if i < len(names) else str(i): round(buf[i] / tot, 4) for i in range(16) if buf[i]}}))
