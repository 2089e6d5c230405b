#!/bin/bash
# SQ / cache counters of the push (or pull) kernels: tools/r5/run_p.sh <tag> <ops> <groups>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
PMC_OPS=$2 PMC_GROUPS=$3 timeout 1500 python tools/pmc_sq.py $1 2.0 > $O/sq.log 2>&1
rm -rf $O/pass*/
cat $O/sq_counters.txt
