set -u
cd ${GRAFT_REPO_ROOT:-.}
for wl in cfg2_1080p_512c_b4 cfg3_4k_1024c_b8; do for v in 0 9; do echo "== $wl variant=$v"; python tools/variant_sweep.py $wl $v 200 2>&1 | grep -v amdgpu.ids | tail -1; done; done
