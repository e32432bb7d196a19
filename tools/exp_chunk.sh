for n in 64 128 512 1024; do
  AB_REPS=4 python tools/lib_ab.py zig_vulkan_amd/libvrt_hip.so zig_vulkan_amd/libvrt_hip_c$n.so cfg4_4k_2048c_b8_sparse V0 V1x 2>&1 | grep -v amdgpu.ids | sed "s/^/chunk $n: /" | cut -c1-50,110-400
done
