#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/s2dprof
S2D_SIGMAS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2dprof -o s2d -- python tools/r5/s2d.py time > gpurun_out/s2dprof.log 2>&1
f=$(ls gpurun_out/s2dprof/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf gpurun_out/s2dprof/*.db gpurun_out/s2dprof/*trace.csv
