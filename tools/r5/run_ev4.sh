#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest2.txt 2>&1; tail -6 $O/pytest2.txt
