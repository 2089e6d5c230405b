#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pre2dprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pre2dprof -o p -- python tools/r5/pre2d_prof.py > gpurun_out/pre2dprof.log 2>&1
f=$(ls gpurun_out/pre2dprof/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r["Name"][:150], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf gpurun_out/pre2dprof/*.db gpurun_out/pre2dprof/*trace.csv
