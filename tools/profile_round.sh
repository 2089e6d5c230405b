#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> pmc     PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) + other configs
#   tools/profile_round.sh <tag> bench   bench line + rocprofv3 --kernel-trace --stats of the same command
# Outputs under gpurun_out/<tag>/; tools/make_profiles.py condenses them into profiles/.
set -u
TAG=${1:-r01}; WHAT=${2:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [ "$WHAT" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/tools/pmc_workload.py > $O/pmc_$c.log 2>&1
  done
  timeout 900 python $R/tools/bench_configs.py > $O/other_configs.json 2> $O/other_configs.err
else
  timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof.err
fi
ls -R $O | head -40
