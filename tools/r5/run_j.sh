#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5j}; mkdir -p $O; shift
cd $R
L=$R/torch-interpol_amd/lib
INTERPOL_HIP_LIB=$L/libinterpol_hip_abl.so timeout 300 python tools/r5/ablate_owner.py 2.0 "$@" 2>&1 | grep sigma | tee $O/ablate.txt
