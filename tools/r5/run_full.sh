#!/bin/bash
# the whole GPU suite + the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5full}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-1500
