#!/bin/bash
# probe_stats for library variants: tools/r5/run_v.sh <tag> "<v...>" cases...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; V=$2; shift; shift
cd $R
L=$R/torch-interpol_amd/lib
for v in $V; do
echo "== $v"; INTERPOL_HIP_LIB=$L/libinterpol_hip$v.so timeout 600 python tools/r5/probe_stats.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee -a $O/stats.txt
done
