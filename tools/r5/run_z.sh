#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5z3}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "shared or third_order or owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
