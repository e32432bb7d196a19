cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
(timeout 600 python tools/flag_check.py 0 8 sparse; timeout 600 python tools/flag_check.py 0 4 sparse;  timeout 300 python tools/flag_check.py 0 8) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dil_flag_check.log
timeout 900 python tools/variant_sweep.py cfg4_4k_2048c_b8_sparse 0/0x400,0,0/0x400,0 2 V0,V1x 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dil_ab.log
