#!/bin/bash
# kernel timeline of the routed pull: with the probe (flags 0) and with the per-tile hand-over alone (AUTO | debug bit 16384)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5x}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for f in 0 $((4 + (16384 << 8))); do
rm -rf $O/st; timeout 600 rocprofv3 --kernel-trace -d $O/st -- python $R/tools/r5/pull_once.py ${2:-0.0} $f > $O/run.log 2>&1
DB=$(find $O/st -name "*.db" | head -1)
echo "== flags $f"; python $R/tools/r5/timeline.py $DB 12 | tee -a $O/timeline.txt
done
rm -rf $O/st
