cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
P=zig_vulkan_amd/ab/libvrt_hip_prev.so; D=zig_vulkan_amd/libvrt_hip_dev.so
{ AB_REPS=9 timeout 300 python tools/lib_ab.py $P $D refapp_1024x576_128x64x128_b4 V0 V1 V2 V1x VG 2>/dev/null
  AB_VARIANT_B=0x500 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $D refapp_1024x576_128x64x128_b4 V0 V1 V2 V1x VG 2>/dev/null
  AB_VARIANT_B=0x600 AB_REPS=9 timeout 300 python tools/lib_ab.py $P $D refapp_1024x576_128x64x128_b4 V0 V1 V2 2>/dev/null
} | tee gpurun_out/r06/ab_bounce_waves.txt | cut -c1-45,60-75,110-140,165-300
