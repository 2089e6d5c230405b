#!/bin/bash
# evidence pass 1: the whole GPU suite, the bench line with its rocprofv3 kernel stats, the other configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
tools/profile_round.sh r05 bench > /dev/null 2>&1; cut -c1-400 $O/bench.json
tools/profile_round.sh r05 other > /dev/null 2>&1; head -c 600 $O/other_configs.json; tail -3 $O/other_configs.err
python tools/make_profiles.py r05 r05 > /dev/null 2>&1
