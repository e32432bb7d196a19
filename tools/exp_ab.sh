# same-box A/B of two builds (or of tuning flags: AB_FLAGS_A / AB_FLAGS_B) over several workloads: exp_ab.sh libA libB
A=$1; B=$2
AB_REPS=300 python tools/lib_ab.py $A $B refapp_1024x576_128x64x128_b4 V0 V1 V2 VG 2>&1 | grep -v amdgpu.ids | cut -c1-45,130-400
AB_REPS=300 python tools/lib_ab.py $A $B cfg2_1080p_512c_b8 V0 V1 V2 VG 2>&1 | grep -v amdgpu.ids | cut -c1-45,130-400
AB_REPS=5 python tools/lib_ab.py $A $B cfg4_4k_2048c_b8_sparse V0 2>&1 | grep -v amdgpu.ids | cut -c1-45,130-400
