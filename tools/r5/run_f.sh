#!/bin/bash
# round-5: magic-format own_accumulate vs the round's base library; push parity subset
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5f}; mkdir -p $O
cd $R
L=$R/torch-interpol_amd/lib
timeout 300 python tools/time_push.py 2.0 0.0 > $O/push_new.txt 2>&1; cat $O/push_new.txt | cut -c1-600
INTERPOL_HIP_LIB=$L/libinterpol_hip_base.so timeout 300 python tools/time_push.py 2.0 0.0 > $O/push_base.txt 2>&1; cat $O/push_base.txt | cut -c1-600
timeout 300 python tools/r5/repro_check.py 2.0 3 > $O/repro.txt 2>&1; cat $O/repro.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
