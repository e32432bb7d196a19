cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
VRT_HIP_LIB=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip_dev.so timeout 600 python tools/experiments/shade1_waves_ab.py cfg3_4k_1024c_b8 7 8 2>/dev/null | tee gpurun_out/r06/shade1_waves8.txt | cut -c1-60,100-260
