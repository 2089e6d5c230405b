#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1500 python tools/fuzz_fastpaths.py 160 ${1:-5} > $O/fuzz_fastpaths.txt 2>&1; tail -6 $O/fuzz_fastpaths.txt | cut -c1-400
timeout 900 python tools/fuzz_scatter_variants.py 240 ${2:-7} > $O/fuzz_scatter.txt 2>&1; tail -6 $O/fuzz_scatter.txt | cut -c1-400
timeout 900 python tools/fuzz_modes.py 120 ${3:-3} > $O/fuzz_modes.txt 2>&1; tail -4 $O/fuzz_modes.txt | cut -c1-400
