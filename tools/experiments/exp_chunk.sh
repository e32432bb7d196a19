# vrt_pool_kernel's chunk of units per atomic: same-box A/B of builds made with
#   make -C zig_vulkan_amd/csrc B=build/c$n OUT_prod=../libvrt_hip_c$n.so EXTRA=-DVRT_POOL_CHUNK=${n}u     (n = 64 128 512 1024)
# against the product library.
for n in 64 128 512 1024; do
  AB_REPS=4 python tools/lib_ab.py zig_vulkan_amd/libvrt_hip.so zig_vulkan_amd/libvrt_hip_c$n.so cfg4_4k_2048c_b8_sparse V0 V1x 2>&1 | grep -v amdgpu.ids | sed "s/^/chunk $n: /"
done
