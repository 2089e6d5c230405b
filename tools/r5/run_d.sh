#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5d}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
INTERPOL_HIP_LIB=$L/libinterpol_hip_stg.so timeout 600 python tools/r5/stagger.py 2.0 > $O/stagger.txt 2>&1; cat $O/stagger.txt
INTERPOL_HIP_LIB=$L/libinterpol_hip_r4.so timeout 300 python tools/r5/repro_check.py 2.0 16 > $O/repro_r4.txt 2>&1; grep -c identical $O/repro_r4.txt; grep n_diff $O/repro_r4.txt | head -2 | cut -c1-300
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "not fuzz" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
