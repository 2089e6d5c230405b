#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5x4}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
for i in 1 2; do for v in "" _bin6; do
INTERPOL_HIP_LIB=$L/libinterpol_hip$v.so timeout 300 python tools/r5/time_owner.py 2.0 0.0 2>&1 | grep lib
done; done
timeout 1500 python -m pytest tests/test_workspace.py tests/test_hip_parity.py -x -q -m gpu -k "history or hand or routed or graph" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
