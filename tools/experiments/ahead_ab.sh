#!/bin/bash
# two-trips-ahead walk loop on cache-resident scenes (development build): path kernel, linear status words, with / without VRT_TUNE_PATH_AHEAD
export VRT_SWEEP_LIB=$PWD/zig_vulkan_amd/libvrt_hip_dev.so
for w in refapp_1024x576_128x64x128_b4 refapp_1024x576_512c_b4; do
  echo "== $w"
  python tools/variant_sweep.py $w 0x800000/0x04,0x800000/0x84,0x800000/0x00,0x200000/0 8 V0,V1,V2 2>&1 | tail -5
done
