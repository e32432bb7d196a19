# same-box A/B of two builds over several workloads: exp_ab.sh libA libB
A=$1; B=$2
AB_REPS=300 python tools/lib_ab.py $A $B refapp_1024x576_128x64x128_b4 V0 V1 V2 VG 2>&1 | grep -v amdgpu.ids
AB_REPS=300 python tools/lib_ab.py $A $B cfg2_1080p_512c_b8 V0 V1 V2 VG 2>&1 | grep -v amdgpu.ids
