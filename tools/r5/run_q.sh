#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5q}; mkdir -p $O; shift
cd $R
timeout 600 python tools/r5/plan_ab.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/plan_ab.txt
