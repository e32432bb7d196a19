for n in 24 32 40 56; do
  for w in cfg2_1080p_512c_b8 cfg1_1080p_256c_b4; do
  AB_REPS=300 python tools/lib_ab.py zig_vulkan_amd/libvrt_hip.so zig_vulkan_amd/libvrt_hip_s$n.so $w V0 V1 V2 2>&1 | grep -v amdgpu.ids | sed "s/^/slots $n: /"
  done
done
