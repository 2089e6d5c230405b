#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5b}; mkdir -p $O
cd $R
timeout 300 python tools/time_push.py 2.0 > $O/push_new.txt 2>&1; cat $O/push_new.txt
INTERPOL_HIP_LIB=$R/torch-interpol_amd/lib/libinterpol_hip_abl.so timeout 300 python tools/r5/ablate_owner.py 2.0 > $O/ablate.txt 2>&1; cat $O/ablate.txt
INTERPOL_HIP_LIB=$R/torch-interpol_amd/lib/libinterpol_hip_prof.so timeout 300 python tools/phase_prof_sorted.py 2.0 push > $O/phase.txt 2>&1; cat $O/phase.txt
KPAT=own_ tools/kstats.sh ${1:-r5b}/ks tools/time_push.py 2.0 > /dev/null 2>&1; grep -A3 "dispatches of" $O/ks/kernel_stats.txt | cut -c1-300
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
