#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5o5c}; mkdir -p $O; shift
cd $R
timeout 1200 python tools/r5/order5_push.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/order5_push.txt
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "orders_4_and_5 or order5 or backward_golden or cfg3 or golden_mid" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
