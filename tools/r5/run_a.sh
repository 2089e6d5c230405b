#!/bin/bash
# round-5 A/B of the owner-computes push: round-4 library vs the working tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5a}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "owner or fold or binned or scatter or push or count" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
INTERPOL_HIP_LIB=$R/torch-interpol_amd/lib/libinterpol_hip_r4.so timeout 300 python tools/time_push.py 2.0 0.0 > $O/push_r4.txt 2>&1
timeout 300 python tools/time_push.py 2.0 0.0 > $O/push_new.txt 2>&1
cat $O/push_r4.txt $O/push_new.txt
INTERPOL_HIP_LIB=$R/torch-interpol_amd/lib/libinterpol_hip_prof.so timeout 300 python tools/phase_prof_sorted.py 2.0 push > $O/phase.txt 2>&1; cat $O/phase.txt
KPAT=own_ tools/kstats.sh ${1:-r5a}/ks tools/time_push.py 2.0 > /dev/null 2>&1; head -30 $O/ks/kernel_stats.txt | cut -c1-150
