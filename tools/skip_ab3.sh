set -u
cd ${GRAFT_REPO_ROOT:-.}
./tools/isa_probe | tail -3
python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/fuzz_parity.py 300 5101 2>&1 | tail -1
python tools/fuzz_parity.py 150 5102 big 2>&1 | tail -1
for wl in cfg2_1080p_512c_b8 cfg3_4k_1024c_b8; do for s in 0 1; do echo "== $wl skip=$s"; VRT_SKIP_TO_BOX=$s python tools/variant_sweep.py $wl 0 200 2>&1 | grep -v amdgpu.ids | tail -1; done; done
VRT_HIP_LIB=$PWD/tools/libvrt_hip_prof.so python tools/frame_phases.py cfg2_1080p_512c_b8 V0,V1,V2,VG 2>&1 | grep -v amdgpu.ids
