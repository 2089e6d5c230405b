#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python tools/r5/order5_push.py 0.0 2.0 3.0 6.0 > $O/order5_push.txt 2>&1; tail -8 $O/order5_push.txt | cut -c1-330
timeout 900 python tools/bench_configs.py 3 > $O/other_configs_3.json 2> $O/other_configs_3.err; cat $O/other_configs_3.json | tr -d '\n' | cut -c1-400
