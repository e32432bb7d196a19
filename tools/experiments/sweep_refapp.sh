export TMPDIR=/tmp
WL=refapp_1024x576_128x64x128_b4
echo "lockstep default:"; python tools/variant_sweep.py $WL 0 200 V0,V1,V2 2>&1 | grep -v amdgpu.ids
export VRT_SWEEP_LIB=$PWD/zig_vulkan_amd/libvrt_hip_dev.so
for FIN in 4 8 16 32; do for BB in 1 2 4 8; do
  V=$(printf "0x%x" $(( (1<<23) | (BB<<24) )))
  echo -n "fin $FIN brick_batch $((BB*4)): "; VRT_DEV_PATH_FIN_BATCH=$FIN python tools/variant_sweep.py $WL $V 200 V0,V1,V2 2>&1 | grep -v amdgpu.ids | cut -c60-
done; done
echo "lockstep 8 waves / path 4,6 waves:"; python tools/variant_sweep.py $WL 0x200805,0x800400,0x800600 200 V0,V1,V2 2>&1 | grep -v amdgpu.ids
