#!/bin/bash
# vrt_path_kernel's batching parameters on the 2048^3 path trace (development build: VRT_DEV_PATH_FIN_BATCH, brick batch in kernel_variant bits 24-27)
export TMPDIR=/tmp VRT_SWEEP_LIB=$PWD/zig_vulkan_amd/libvrt_hip_dev.so
WL=cfg4_4k_2048c_b8_sparse
for FIN in ${FINS:-16 32 48 64}; do
  echo -n "fin $FIN: "; VRT_DEV_PATH_FIN_BATCH=$FIN python tools/variant_sweep.py $WL ${VARS:-0x800500,0xa800500,0xc800500,0xe800500} 2 V0 2>&1 | grep -v amdgpu.ids | awk '{printf "%s %s %s | ", $2, $(NF-1), $NF} END {print ""}'
done
