for n in 28 52 64 96; do
  AB_REPS=200 python tools/lib_ab.py zig_vulkan_amd/libvrt_hip.so zig_vulkan_amd/libvrt_hip_s$n.so refapp_1024x576_128x64x128_b4 V0 V1 V2 2>&1 | grep -v amdgpu.ids | sed "s/^/slots $n: /" | cut -c1-50,140-330
done
