#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> bench   bench line + rocprofv3 --kernel-trace --stats of the same command
#   tools/profile_round.sh <tag> pmc     PMC passes FETCH_SIZE / WRITE_SIZE (separate runs) at sigma = 2 and at the identity
#                                        (calibration: there the HBM traffic IS the algorithmic byte count)
#   tools/profile_round.sh <tag> sq      SQ / TCP counters of the headline kernels and of the 2-D kernels (tools/pmc_sq.py)
#   tools/profile_round.sh <tag> other   the other BASELINE configs, follow-ups, roughness sweep (tools/bench_configs.py)
#   tools/profile_round.sh <tag> phase   phase shares (needs lib/libinterpol_hip_prof.so: make PROF=1 BUILD=build_prof LIB=...)
#   tools/profile_round.sh <tag> micro   LDS microbenchmarks (tools/microbench)
# Outputs under gpurun_out/<tag>/; tools/make_profiles.py condenses them into profiles/.
set -u
TAG=${1:-r03}; WHAT=${2:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
case $WHAT in
pmc)
  for s in 2.0 0.0; do for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_s$s -- python $R/tools/pmc_workload.py $s > $O/pmc_${c}_s$s.log 2>&1
  done; done
  # raw per-dispatch counters of THESE passes -> $O/pmc_raw.json (the databases stay on the box; make_profiles.py builds the tables from it)
  python $R/tools/make_profiles.py $TAG ${ROUND:-r03} > /dev/null 2>&1
  rm -rf $O/pmc_*_s*/ ;;
sq)
  PMC_OPS=pull,push timeout 1500 python $R/tools/pmc_sq.py $TAG/sq_cfg2 2.0 > $O/sq_cfg2.log 2>&1
  PMC_OPS=pull2d,push2d PMC_GROUPS=0,1,2,3,6,7,10,11,12,13 timeout 1500 python $R/tools/pmc_sq.py $TAG/sq_cfg5 2.0 > $O/sq_cfg5.log 2>&1 ;;
other)
  timeout 1500 python $R/tools/bench_configs.py 1 3 4 5 f b r > $O/other_configs.json 2> $O/other_configs.err ;;
phase)
  export INTERPOL_HIP_LIB=$R/torch-interpol_amd/lib/libinterpol_hip_prof.so
  for s in 2.0 0.0; do
    timeout 300 python $R/tools/phase_prof_sorted.py $s pull >> $O/phase_split.txt 2>&1
    timeout 300 python $R/tools/phase_prof.py $s >> $O/phase_split.txt 2>&1
  done
  unset INTERPOL_HIP_LIB                      # the 2-D kernels: ablation timings of the production library (the
  for op in pull push; do                     # phase marks add barriers that distort kernels this short)
    echo "# tools/ablate_2d.py $op: ms per call at config 5 with phases disabled (pull bits: 1 staging, 2 taps, 4 stores, 8 coordinate loads; push bits: 1 taps, 2 flush)" >> $O/phase_split.txt
    timeout 300 python $R/tools/ablate_2d.py $op >> $O/phase_split.txt 2>&1
  done ;;
micro)
  for m in lds_gather lds_atomics; do
    hipcc -O3 --offload-arch=gfx950 $R/tools/microbench/$m.hip -o /tmp/$m.bin && timeout 120 /tmp/$m.bin > $O/micro_$m.txt 2>&1
  done ;;
*)
  timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof.err ;;
esac
ls $O | head -40
