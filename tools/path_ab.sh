set -u
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "path or cfg4 or bounce" 2>&1 | grep -E "passed|failed|error" | tail -2
python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "memory_layouts" 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 900 python tools/fuzz_parity.py 40 9101 pow2 2>&1 | tail -1
FRAMES=3 bash tools/ab_libs.sh tools/libvrt_hip_base.so zig_vulkan_amd/libvrt_hip.so cfg4_4k_2048c_b8_sparse V0,V1x
