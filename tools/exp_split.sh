for L in libvrt_hip.so libvrt_hip_exp1.so libvrt_hip_exp2.so libvrt_hip_exp3.so; do
  echo "== $L"; VRT_SWEEP_LIB=$PWD/zig_vulkan_amd/$L python tools/variant_sweep.py refapp_1024x576_128x64x128_b4 0,0x30000 200 V0,V1,V2,VG 2>&1 | grep -v amdgpu.ids
done
