#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5t}; mkdir -p $O; shift
cd $R
timeout 900 python tools/r5/probe_stats.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/probe_stats.txt
