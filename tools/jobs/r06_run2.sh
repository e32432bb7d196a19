set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export VRT_HIP_LIB=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip_dev.so
for s in "2:1" "4:1" "8:1" "4:2" "8:2" "16:1"; do
  echo "## VRT_DEV_SHARD_SPLIT=$s"
  VRT_DEV_SHARD_SPLIT=$s timeout 300 python tools/experiments/literal_leg_probe.py 8 2>/dev/null | grep -v "^#"
done > gpurun_out/r06/shard_split_probe.txt
cat gpurun_out/r06/shard_split_probe.txt
for s in "2:1" "4:1" "8:1"; do
  echo "## VRT_DEV_SHARD_SPLIT=$s world 4"
  VRT_DEV_SHARD_SPLIT=$s timeout 300 python tools/experiments/literal_leg_probe.py 4 2>/dev/null | grep -v "^#"
done > gpurun_out/r06/shard_split_probe_w4.txt
cat gpurun_out/r06/shard_split_probe_w4.txt
