#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/restrictprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/restrictprof -o p -- python tools/r5/restrict_prof.py > gpurun_out/restrictprof.log 2>&1
f=$(ls gpurun_out/restrictprof/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r["Name"][:150], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf gpurun_out/restrictprof/*.db gpurun_out/restrictprof/*trace.csv
