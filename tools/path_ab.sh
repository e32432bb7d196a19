set -u
cd ${GRAFT_REPO_ROOT:-.}
for b in 16 24 48 64; do echo "== fin_batch=$b"; VRT_PATH_FIN_BATCH=$b python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
for b in 4 6 12 16; do echo "== brick_batch=$((b*4))"; python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse $((b<<24)) 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
for g in 1024 1536 3072; do echo "== groups=$g"; VRT_PATH_GROUPS=$g python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "== 4 waves"; python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse $(( (4<<8) | (1<<23) )) 3 V0 2>&1 | grep -v amdgpu.ids | tail -1
