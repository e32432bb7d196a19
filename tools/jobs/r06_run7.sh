set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
P=zig_vulkan_amd/ab/libvrt_hip_prev.so; N=zig_vulkan_amd/libvrt_hip.so
{ AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N refapp_1024x576_128x64x128_b4 V0 V1 V2 V1x VG 2>/dev/null
  AB_REPS=9 timeout 300 python tools/lib_ab.py $N $P refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
  AB_SPP=1 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $N refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
  AB_WIDTH=1920 AB_HEIGHT=1080 AB_REPS=5 timeout 300 python tools/lib_ab.py $P $N refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
} > gpurun_out/r06/ab_overlap.txt; cat gpurun_out/r06/ab_overlap.txt
timeout 900 python -m pytest tests/test_reference_parity_gpu.py tests/test_parity_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r06/t_7.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06/t_7.log
