cd ${GRAFT_REPO_ROOT:-.}
A=zig_vulkan_amd/libvrt_hip.so; B=zig_vulkan_amd/libvrt_hip_early.so
FRAMES=2 bash tools/ab_libs.sh $A $B cfg4_4k_2048c_b8_sparse V0,V1x
VRT_HIP_LIB=$PWD/$B python tools/flag_check.py 0 8 sparse 2>&1 | grep -v amdgpu | tail -3
VRT_HIP_LIB=$PWD/$B python tools/flag_check.py 0 8 2>&1 | grep -v amdgpu | tail -2
