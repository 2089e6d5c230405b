#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5s}; mkdir -p $O; T=$1; shift
cd $R
timeout 600 python tools/r5/plan_ab.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/plan_ab.txt
KPAT=own_ tools/kstats.sh $T/ks tools/r5/time_owner.py 2.0 > /dev/null 2>&1; grep -A2 "dispatches of" $O/ks/kernel_stats.txt | cut -c1-300; head -8 $O/ks/kernel_stats.txt | cut -c1-60,120-170
