#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5t5}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "orders_4_and_5 or order5 or backward_golden or cfg3 or golden_mid" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
