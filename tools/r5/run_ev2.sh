#!/bin/bash
# evidence pass 2: PMC traffic, SQ counters of the headline kernels, config 3 / 4 rows again (scatter5, shared bricks), rough rows, tolerance report
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
ROUND=r05 tools/profile_round.sh r05 pmc > /dev/null 2>&1; ls $O | head -30
PMC_OPS=pull,push PMC_GROUPS=0,1,2,3,4,7,9,10,11 timeout 1500 python tools/pmc_sq.py r05/sq_cfg2 2.0 > $O/sq_cfg2.log 2>&1; rm -rf $O/sq_cfg2/pass*/
timeout 900 python tools/bench_configs.py 3 4 > $O/other_configs_34.json 2> $O/other_configs_34.err; head -c 900 $O/other_configs_34.json
timeout 900 python tools/rough_rows.py > $O/rough_rows.txt 2>&1; cut -c1-200 $O/rough_rows.txt | tail -7
timeout 900 python tools/r5/order5_push.py 0.0 2.0 3.0 6.0 > $O/order5_push.txt 2>&1; tail -8 $O/order5_push.txt | cut -c1-250
timeout 900 python tools/r5/cfg4.py 8 16 32 64 > $O/cfg4.txt 2>&1; tail -4 $O/cfg4.txt
