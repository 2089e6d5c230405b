#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5w2}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_workspace.py tests/test_hip_parity.py -x -q -m gpu -k "workspace or routed or graph or pull or f64 or float64 or handback or hand_back or history" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
