#!/bin/bash
# A/B of library variants lib/libinterpol_hip<v>.so with tools/r5/time_both.py: tools/r5/run_ab.sh <tag> <v...>   ("" = the default library)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
L=$R/torch-interpol_amd/lib
for i in 1 2; do
for v in "$@"; do
INTERPOL_HIP_LIB=$L/libinterpol_hip$v.so timeout 300 python tools/r5/time_both.py 2.0 0.0 2>&1 | grep lib | tee -a $O/ab.txt
done; done
