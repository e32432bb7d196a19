cd ${GRAFT_REPO_ROOT:-.}
A=zig_vulkan_amd/libvrt_hip.so; B=zig_vulkan_amd/libvrt_hip_early.so
bash tools/ab_libs.sh $A $B cfg2_1080p_512c_b8 V0,V1,V2
FRAMES=60 bash tools/ab_libs.sh $A $B cfg3_4k_1024c_b8 V0,V1
FRAMES=100 bash tools/ab_libs.sh $A $B refapp_1024x576_128x64x128_b4 V0,V2
FRAMES=300 bash tools/ab_libs.sh $A $B cfg1_1080p_256c_b4 V0,V1
FRAMES=2 bash tools/ab_libs.sh $A $B cfg4_4k_2048c_b8_sparse V0
for b in 8 4; do VRT_HIP_LIB=$PWD/$B python tools/flag_check.py 0 $b 2>&1 | grep -v amdgpu | tail -2; done
VRT_HIP_LIB=$PWD/$B timeout 600 python tools/fuzz_parity.py 300 5 2>&1 | tail -1 | cut -c1-100
