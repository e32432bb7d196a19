set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/skip_ab3
python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/fuzz_parity.py 200 4203 big 2>&1 | tail -2
python tools/fuzz_parity.py 300 4204 2>&1 | tail -2
for s in 0 1; do echo "== cfg4 skip=$s"; VRT_SKIP_TO_BOX=$s python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0 3 V0,V1,V1x 2>&1 | grep -v amdgpu.ids | tail -1; done
for s in 0 1; do echo "== refapp skip=$s"; VRT_SKIP_TO_BOX=$s python tools/variant_sweep.py refapp_1024x576_512c_b4 0 50 2>&1 | grep -v amdgpu.ids | tail -1; done
for s in 0 1; do echo "== cfg1 skip=$s"; VRT_SKIP_TO_BOX=$s python tools/variant_sweep.py cfg1_1080p_256c_b4 0 200 2>&1 | grep -v amdgpu.ids | tail -1; done
for s in 0 1; do echo "== cfg0 skip=$s"; VRT_SKIP_TO_BOX=$s python tools/variant_sweep.py cfg0_256x256_64c_b4 0 200 2>&1 | grep -v amdgpu.ids | tail -1; done
