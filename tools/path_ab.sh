set -u
cd ${GRAFT_REPO_ROOT:-.}
for e in 0 1; do echo "== cfg4 eager_start=$e"; VRT_PATH_EAGER_START=$e python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0,V1x 2>&1 | grep -v amdgpu.ids | tail -1; done
for b in 16 32 48; do echo "== cfg4 fin_batch=$b"; VRT_PATH_FIN_BATCH=$b python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
for b in 4 8 12 ; do echo "== cfg4 brick_batch=$((b*4))"; python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse $((b<<24)) 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
