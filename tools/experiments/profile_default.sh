#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the plain default command `python bench.py` (two frames in flight).
# usage: tools/experiments/profile_default.sh <tag>
set -u
TAG=${1:-default}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --no-cpu-baseline --pmc off > $OUT/stats.log 2>&1
cd $ROOT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
