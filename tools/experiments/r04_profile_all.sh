#!/bin/bash
# Runs on the GPU box (via gpurun): for every BASELINE workload a plain bench line (with the live PMC leg and the CPU
# baselines) and the rocprofv3 --kernel-trace --stats summary of the same command without the CPU / PMC legs.
# usage: tools/experiments/r03_profile_all.sh <tag> [workload ...]
set -u
TAG=${1:-r04}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${*:-"cfg0_256x256_64c_b4 cfg1_1080p_256c_b4 cfg2_1080p_512c_b8 cfg2_1080p_512c_b4 cfg3_4k_1024c_b8 cfg4_4k_2048c_b8_sparse refapp_1024x576_128x64x128_b4"}
OUTROOT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUTROOT
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  OUT=$OUTROOT/$W
  mkdir -p $OUT
  python $ROOT/bench.py --workload $W > $OUT/bench.json 2> $OUT/bench.err
  echo "$W bench rc=$? $(head -c 300 $OUT/bench.json)"
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --workload $W --no-cpu-baseline --pmc off > $OUT/stats.log 2>&1
  python $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  tail -12 $OUT/summary.txt
  rm -rf $OUT/stats   # the rocpd databases are large; the summary is what is kept
done
