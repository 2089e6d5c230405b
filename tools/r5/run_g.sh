#!/bin/bash
# round-5: ablations and phase shares of own_accumulate (magic format)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5g}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
INTERPOL_HIP_LIB=$L/libinterpol_hip_abl.so timeout 300 python tools/r5/ablate_owner.py 2.0 > $O/ablate.txt 2>&1; cat $O/ablate.txt
INTERPOL_HIP_LIB=$L/libinterpol_hip_prof.so timeout 300 python tools/phase_prof_sorted.py 2.0 push > $O/phase.txt 2>&1; cat $O/phase.txt
KPAT=own_ tools/kstats.sh ${1:-r5g}/ks tools/time_push.py 2.0 > /dev/null 2>&1; grep -B2 -A12 "dispatches of" $O/ks/kernel_stats.txt | cut -c1-200
