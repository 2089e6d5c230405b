#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5c4}; mkdir -p $O; shift
cd $R
timeout 900 python tools/r5/cfg4.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/cfg4.txt
