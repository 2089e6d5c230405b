#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5r}; mkdir -p $O
cd $R
KPAT=own_ tools/kstats.sh ${1:-r5r}/ks tools/r5/time_owner.py 2.0 > /dev/null 2>&1; grep -A8 "dispatches of" $O/ks/kernel_stats.txt | cut -c1-700; head -20 $O/ks/kernel_stats.txt | cut -c1-160
