#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5e}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
INTERPOL_HIP_LIB=$L/libinterpol_hip_stg.so timeout 600 python tools/r5/binstores.py > $O/binstores.txt 2>&1; cat $O/binstores.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
