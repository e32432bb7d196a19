#!/bin/bash
# Runs on the GPU box (via gpurun): the reference's own benchmark protocol (tools/flythrough.py) on the app's default run and on the
# headline workload, and the rocprofv3 --kernel-trace --stats summary of the app's run (lists vrt_denoise_kernel beside the trace kernel).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_fly
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/flythrough.py refapp_1024x576_128x64x128_b4 30 --out $OUT/r04_flythrough_refapp.json 2> $OUT/refapp.err | tail -1
python $ROOT/tools/flythrough.py cfg2_1080p_512c_b8 30 --out $OUT/r04_flythrough_headline.json 2> $OUT/headline.err | tail -1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/tools/flythrough.py refapp_1024x576_128x64x128_b4 30 > $OUT/stats.log 2>&1
python $ROOT/tools/summarize_prof.py $OUT > $OUT/r04_flythrough_refapp_rocprof.txt 2>&1
head -40 $OUT/r04_flythrough_refapp_rocprof.txt
rm -rf $OUT/stats
