#!/bin/bash
# Runs on the GPU box (via gpurun): the round-4 evidence files that are not bench lines (profiles/README.md, round 4).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_evidence
mkdir -p $OUT
cd $ROOT && export TMPDIR=/tmp
WL=cfg4_4k_2048c_b8_sparse
APP=refapp_1024x576_128x64x128_b4
VRT_HIP_LIB=$ROOT/zig_vulkan_amd/libvrt_hip_prof.so timeout 400 python tools/path_profile.py $WL V0 2>&1 | grep -v amdgpu > $OUT/r04_cfg4_phase_profile_pool.txt
timeout 600 tools/pmc_path.sh 0 $WL 0 pool 2>&1 | grep -E "^#|pool_kernel" > $OUT/r04_cfg4_pool_counters.txt
timeout 400 python tools/experiments/pool_ab.py $WL V0 V1x 2>&1 | grep -v amdgpu > $OUT/r04_pool_ab.txt
timeout 900 python tools/experiments/pool_sweep.py $WL V0 16:48:48:32,16:56:56:24,8:48:48:32,24:48:48:32,16:40:40:32,16:62:62:32,16:48:48:8 5:64:2,4:64:4,5:64:1,5:40:4,6:56:1 2>&1 | grep -v amdgpu > $OUT/r04_pool_sweep.txt
timeout 120 tools/ubench/trip_bench > $OUT/r04_trip_bench.txt 2>&1
timeout 300 python tools/experiments/timeline.py 0 $APP 2>&1 | grep -v amdgpu > $OUT/r04_refapp_timeline.txt
timeout 600 python tools/experiments/slow_waves.py V0 $APP 2>&1 | grep -v amdgpu > $OUT/r04_refapp_slow_waves.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r04_bench_driver_protocol_steps20.json 2> $OUT/driver.err
tools/experiments/r04_flythrough.sh > $OUT/fly.log 2>&1
python tools/experiments/small_frame_ab.py cfg0_256x256_64c_b4 2>&1 | grep -v amdgpu > $OUT/r04_small_frame_ab.txt
python tools/experiments/timeline_tail.py $APP V0,V1,V2,VG 2>&1 | grep -v amdgpu > $OUT/r04_timeline_tail_refapp.txt
{ python tools/experiments/timeline_tail.py cfg2_1080p_512c_b8; python tools/experiments/timeline_tail.py cfg0_256x256_64c_b4; } 2>&1 | grep -v amdgpu > $OUT/r04_timeline_tail_headline.txt
python tools/experiments/timeline_order.py cfg2_1080p_512c_b8 2>&1 | grep -v amdgpu > $OUT/r04_timeline_order_headline.txt
python tools/experiments/timeline_order.py $APP VG,V1 2>&1 | grep -v amdgpu > $OUT/r04_timeline_order_refapp.txt
python tools/experiments/sky_frame.py 2>&1 | grep -v amdgpu > $OUT/r04_sky_frame.txt
ls -la $OUT
