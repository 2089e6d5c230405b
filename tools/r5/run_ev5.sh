#!/bin/bash
# round-5 evidence of the 2-D bricks: config 5's rows (tools/bench_configs.py 5) and the sigma sweeps of the three organisations
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5ev5; mkdir -p $O; cd $R
timeout 900 python tools/bench_configs.py 5 > $O/other_configs_5.json 2> $O/other_configs_5.err; head -c 600 $O/other_configs_5.json
S2D_SIGMAS=0,0.5,2,4,6,8,16 timeout 900 python tools/r5/s2d.py time > $O/s2d_time.txt 2>&1; tail -7 $O/s2d_time.txt
S2D_SIGMAS=0,2,4,6,8,16 timeout 900 python tools/r5/g2d.py time > $O/g2d_time.txt 2>&1; tail -6 $O/g2d_time.txt
