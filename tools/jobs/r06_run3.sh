set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
P=zig_vulkan_amd/ab/libvrt_hip_prev.so; N=zig_vulkan_amd/libvrt_hip.so
{ for i in 1 2; do AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg3_4k_1024c_b8 V0 V1 V2 V1x VG 2>/dev/null; done
  AB_REPS=9 timeout 300 python tools/lib_ab.py $N $P cfg3_4k_1024c_b8 V0 V1 V2 2>/dev/null
  AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
  AB_REPS=15 timeout 300 python tools/lib_ab.py $P $N cfg2_1080p_512c_b8 V0 V1 V2 2>/dev/null
  AB_SPP=4 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N cfg3_4k_1024c_b8 V1 V2 2>/dev/null
} > gpurun_out/r06/ab_fresh_lane.txt; cat gpurun_out/r06/ab_fresh_lane.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/t_all3.log 2>&1; echo "all gpu tests rc=$?"; tail -5 gpurun_out/r06/t_all3.log
