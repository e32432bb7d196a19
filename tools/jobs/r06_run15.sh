set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
P=zig_vulkan_amd/ab/libvrt_hip_prev.so; N=zig_vulkan_amd/libvrt_hip.so
{ AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg3_4k_1024c_b8 V0 V1 V2 V1x VG 2>/dev/null
  AB_REPS=9 timeout 300 python tools/lib_ab.py $N $P cfg3_4k_1024c_b8 V0 V1 V2 2>/dev/null
  AB_SPP=4 AB_REPS=7 timeout 300 python tools/lib_ab.py $P $N cfg3_4k_1024c_b8 V1 V2 2>/dev/null
  AB_SPP=2 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b4 V0 V1 V2 2>/dev/null
  AB_SPP=2 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b8 V0 V1 V2 2>/dev/null
  AB_SPP=2 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg1_1080p_256c_b4 V0 V1 V2 2>/dev/null
  AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b8 V0 V1 V2 VG 2>/dev/null
  AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
} > gpurun_out/r06/ab_shade1_7waves.txt; cut -c1-60,90-130,165-330 gpurun_out/r06/ab_shade1_7waves.txt
for L in $P $N $P $N; do VRT_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$L headline', j['value'], j['ms_per_step'])"; done | tee gpurun_out/r06/bench_ab_7w.txt
for L in $P $N $P $N; do VRT_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --workload cfg3_4k_1024c_b8 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$L cfg3', j['value'], j['ms_per_step'], j['roofline']['kernel'])"; done | tee -a gpurun_out/r06/bench_ab_7w.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/t_7w.log 2>&1; echo "gpu tests rc=$?"; grep -a "passed\|failed" gpurun_out/r06/t_7w.log | tail -1
