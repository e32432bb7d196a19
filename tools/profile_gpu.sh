#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --pmc off --frames-in-flight 1 $*"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
# PMC passes, each in its own run, kernel-trace only
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
done
cd $ROOT
python tools/summarize_prof.py $OUT $OUT/pmc.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/stats $OUT/pmc[0-9]   # the rocpd databases are large (gpurun copies back at most 64 MiB); the summaries are what is kept
