set -u
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "path or cfg4 or bounce" 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 900 python tools/fuzz_parity.py 60 8101 pow2 2>&1 | tail -2
for l in 0 1; do echo "== cfg4 halfblocks=$l"; VRT_PATH_HALFBLOCKS=$l python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0,V1,V1x 2>&1 | grep -v amdgpu.ids | tail -1; done
