#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5h}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
for i in 1 2; do
timeout 300 python tools/time_push.py 2.0 0.0 2>&1 | grep sigma | cut -c1-420
INTERPOL_HIP_LIB=$L/libinterpol_hip_base.so timeout 300 python tools/time_push.py 2.0 2>&1 | grep sigma | cut -c1-420
done
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
