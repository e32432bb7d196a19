cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "probe_child" 2>&1 | tail -2
VRT_HIP_LIB=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip_dev.so timeout 600 python tools/experiments/shade1_waves_ab.py cfg3_4k_1024c_b8 7 5 2>/dev/null | tee gpurun_out/r06/shade1_waves.txt | cut -c1-260
VRT_HIP_LIB=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip_dev.so timeout 600 python tools/experiments/shade1_waves_ab.py cfg3_4k_1024c_b8 7 2>/dev/null | tee -a gpurun_out/r06/shade1_waves.txt | cut -c1-260
