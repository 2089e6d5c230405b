#!/bin/bash
# A/B of library variants + the push parity subset on the default library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
L=$R/torch-interpol_amd/lib
for i in 1 2; do
for v in "$@"; do
INTERPOL_HIP_LIB=$L/libinterpol_hip$v.so timeout 300 python tools/r5/time_owner.py 2.0 0.0 2>&1 | grep lib | tee -a $O/ab.txt
done; done
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
