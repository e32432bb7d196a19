cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
A=zig_vulkan_amd/libvrt_hip.so; B=zig_vulkan_amd/ab/libvrt_hip_mid.so
{ AB_REPS=5 timeout 600 python tools/lib_ab.py $A $B cfg4_4k_2048c_b8_sparse V0 V1 V1x 2>/dev/null
  AB_REPS=5 timeout 600 python tools/lib_ab.py $B $A cfg4_4k_2048c_b8_sparse V0 V1x 2>/dev/null; } | tee gpurun_out/r06/ab_midcheck.txt | cut -c1-300
