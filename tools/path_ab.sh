set -u
cd ${GRAFT_REPO_ROOT:-.}
VRT_HIP_LIB=$PWD/tools/libvrt_hip_prof.so python tools/path_profile.py cfg4_4k_2048c_b8_sparse V0 2>&1 | grep -v amdgpu.ids
for r in 2 4 16 32; do echo "== skip_rounds=$r"; VRT_PATH_SKIP_ROUNDS=$r python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
for r in 8 16 48 64; do echo "== ready_batch=$r"; VRT_PATH_READY_BATCH=$r python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
