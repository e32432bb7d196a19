#!/bin/bash
# Runs on the GPU box (via gpurun): the evidence files of a round, all from ONE build.
#   part "bench":    for every BASELINE workload (+ the reference app's own run) the plain `python bench.py --workload W` line (live PMC leg,
#                    CPU baselines) and the `rocprofv3 --kernel-trace --stats` summary of the same command without the CPU / PMC legs;
#   part "driver":   `python bench.py --steps 20 --warmup 5` — what the driver runs;
#   part "fly":      the reference's own benchmark protocol (tools/flythrough.py) on the app's run and on the headline workload;
#   part "cfg4":     the 2048^3 path trace's phase profile (make prof) and its fabric traffic by kernel (tools/pmc_traffic.sh);
#   part "soak":     determinism soak (tools/soak.py).
#   part "fuzz":     randomised parity fuzz against the oracle (tools/fuzz_parity.py): general, big frames, pow2, pool.
# usage: tools/evidence.sh <tag> [part ...]        (default: every part)      -> gpurun_out/evidence_<tag>/
set -u
TAG=${1:-r05}; shift || true
PARTS=${*:-"bench driver fly cfg4 soak fuzz"}
FAILED=""
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/evidence_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
WLS=${WLS:-"cfg0_256x256_64c_b4 cfg1_1080p_256c_b4 cfg2_1080p_512c_b8 cfg2_1080p_512c_b4 cfg3_4k_1024c_b8 cfg4_4k_2048c_b8_sparse refapp_1024x576_128x64x128_b4"}
APP=refapp_1024x576_128x64x128_b4
for part in $PARTS; do
  case $part in
  bench)
    cd /tmp
    for W in $WLS; do
      python $ROOT/bench.py --workload $W > $OUT/${TAG}_bench_$W.json 2> $OUT/bench_$W.err
      echo "$W bench rc=$? $(head -c 200 $OUT/${TAG}_bench_$W.json)"
      mkdir -p $OUT/$W
      rocprofv3 --kernel-trace --stats -d $OUT/$W/stats -o stats -- python $ROOT/bench.py --workload $W --no-cpu-baseline --pmc off > $OUT/$W/bench.json 2> $OUT/$W/stats.log
      python $ROOT/tools/summarize_prof.py $OUT/$W --bench-line $OUT/$W/bench.json --require-phases > $OUT/${TAG}_$W.txt 2>&1 \
        || { echo "!! $W: summarize_prof.py could not split the launches by bench.py phase (see $OUT/${TAG}_$W.txt)"; cp $OUT/$W/bench.json $OUT/failed_bench_$W.json; cp $OUT/$W/launch_durations.json $OUT/failed_durations_$W.json 2>/dev/null; FAILED="$FAILED $W"; }
      rm -rf $OUT/$W   # the rocpd databases are large; the summary is what is kept
    done ;;
  driver)
    cd $ROOT && python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_protocol_steps20.json 2> $OUT/driver.err; head -c 200 $OUT/${TAG}_bench_driver_protocol_steps20.json; echo ;;
  fly)
    cd /tmp
    python $ROOT/tools/flythrough.py $APP 30 --out $OUT/${TAG}_flythrough_refapp.json 2> $OUT/fly_refapp.err | tail -1
    python $ROOT/tools/flythrough.py cfg2_1080p_512c_b8 30 --out $OUT/${TAG}_flythrough_headline.json 2> $OUT/fly_headline.err | tail -1 ;;
  cfg4)
    cd $ROOT
    VRT_HIP_LIB=$ROOT/zig_vulkan_amd/libvrt_hip_prof.so timeout 400 python tools/path_profile.py cfg4_4k_2048c_b8_sparse V0 2>&1 | grep -v amdgpu > $OUT/${TAG}_cfg4_phase_profile_pool.txt
    timeout 600 bash tools/pmc_traffic.sh 0 cfg4_4k_2048c_b8_sparse V0 $TAG 2>&1 | grep -E "^#|pool_kernel|resolve" > $OUT/${TAG}_cfg4_pmc_traffic.txt
    cat $OUT/${TAG}_cfg4_phase_profile_pool.txt ;;
  soak)
    cd $ROOT
    { timeout 600 python tools/soak.py 1500; timeout 600 python tools/soak.py 600 $APP 1; timeout 600 python tools/soak.py 600 $APP 2; timeout 600 python tools/soak.py 300 cfg3_4k_1024c_b8 2;
      timeout 900 python tools/soak.py 45 cfg4_4k_2048c_b8_sparse 2; } 2>&1 | grep -v amdgpu > $OUT/${TAG}_soak.txt; cat $OUT/${TAG}_soak.txt | cut -c1-120 ;;
  fuzz)
    cd $ROOT
    { for seed in 6101 6102 6103; do timeout 900 python tools/fuzz_parity.py 1000 $seed 2>&1 | tail -1; done
      timeout 900 python tools/fuzz_parity.py 400 6201 big 2>&1 | tail -1
      timeout 900 python tools/fuzz_parity.py 1000 6301 pow2 2>&1 | tail -1
      for seed in 6401 6402; do timeout 900 python tools/fuzz_parity.py 1000 $seed pool 2>&1 | tail -1; done; } | grep -v amdgpu > $OUT/${TAG}_fuzz.txt; cat $OUT/${TAG}_fuzz.txt | cut -c1-200 ;;
  esac
done
ls -la $OUT
if [ -n "$FAILED" ]; then echo "!! summaries WITHOUT the per-phase split:$FAILED"; exit 3; fi
