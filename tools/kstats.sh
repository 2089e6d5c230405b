#!/bin/bash
# tools/kstats.sh <tag> <python script + args...>: rocprofv3 --kernel-trace --stats of a workload, per-kernel summary
# printed and kept under gpurun_out/<tag>/.  Run through gpurun from the repo root.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
S=$1; shift
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/$S "$@" > $O/run.log 2>&1
tail -5 $O/run.log
DB=$(find $O/stats -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB > $O/kernel_stats.txt
[ -n "${KPAT:-}" ] && python $R/tools/prof_summary.py --dispatches $KPAT $DB >> $O/kernel_stats.txt
rm -rf $O/stats
cut -c1-90,120-200 $O/kernel_stats.txt
