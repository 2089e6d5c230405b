#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5o5}; mkdir -p $O; shift
cd $R
timeout 1200 python tools/r5/order5_push.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/order5_push.txt
