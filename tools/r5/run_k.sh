#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5k}; mkdir -p $O; shift
cd $R
L=$R/torch-interpol_amd/lib
for d in "$@"; do
INTERPOL_HIP_LIB=$L/libinterpol_hip_prof.so timeout 300 python tools/phase_prof_sorted.py 2.0 push $d 2>&1 | grep share | tee -a $O/phase.txt
done
