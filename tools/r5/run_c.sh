#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5c}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
timeout 300 python tools/r5/repro_check.py 2.0 10 > $O/repro.txt 2>&1; cat $O/repro.txt | cut -c1-400
for v in prof profv1; do INTERPOL_HIP_LIB=$L/libinterpol_hip_$v.so timeout 300 python tools/phase_prof_sorted.py 2.0 push > $O/phase_$v.txt 2>&1; cat $O/phase_$v.txt; done
for i in 1 2; do
timeout 300 python tools/time_push.py 2.0 2>&1 | grep sigma | cut -c1-330
INTERPOL_HIP_LIB=$L/libinterpol_hip_v1.so timeout 300 python tools/time_push.py 2.0 2>&1 | grep sigma | cut -c1-330
done
PMC_OPS=push PMC_GROUPS=0,1,2,3,7 timeout 900 python tools/pmc_sq.py ${1:-r5c}/sq_v0 2.0 > $O/sq_v0.log 2>&1
INTERPOL_HIP_LIB=$L/libinterpol_hip_v1.so PMC_OPS=push PMC_GROUPS=0,1,2,3,7 timeout 900 python tools/pmc_sq.py ${1:-r5c}/sq_v1 2.0 > $O/sq_v1.log 2>&1
for v in v0 v1; do echo == $v; grep -A28 "own_accumulate" $O/sq_$v/sq_counters.txt | head -32; done
rm -rf $O/sq_v0/pass* $O/sq_v1/pass*
