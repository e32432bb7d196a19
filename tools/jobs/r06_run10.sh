set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
P=zig_vulkan_amd/ab/libvrt_hip_prev.so; N=zig_vulkan_amd/libvrt_hip.so
{ for i in 1 2; do AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b8 V0 V1 V2 V1x VG 2>/dev/null; done
  AB_REPS=15 timeout 300 python tools/lib_ab.py $N $P cfg2_1080p_512c_b8 V0 V1 V2 2>/dev/null
  AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg1_1080p_256c_b4 V0 V1 V2 VG 2>/dev/null
  AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b4 V0 V1 V2 2>/dev/null
  AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg0_256x256_64c_b4 V0 V1 V2 2>/dev/null
} > gpurun_out/r06/ab_fresh_lane_shade2.txt; cat gpurun_out/r06/ab_fresh_lane_shade2.txt | cut -c1-330
for L in $P $N $P $N; do VRT_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$L', j['value'], j['ms_per_step'], j['ms_per_step_single_stream'], j['roofline']['kernel_ms_per_view'])"; done | tee gpurun_out/r06/bench_ab_shade2.txt
for L in $P $N $P $N; do VRT_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --no-cpu-baseline --pmc off --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$L driver protocol', j['value'], j['ms_per_step'])"; done | tee -a gpurun_out/r06/bench_ab_shade2.txt
