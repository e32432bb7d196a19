# same-box A/B of two builds (or of tuning flags: AB_FLAGS_A / AB_FLAGS_B) over several workloads: exp_ab.sh libA libB [workload ...]
A=$1; B=$2; shift; shift
for w in ${*:-cfg2_1080p_512c_b8 cfg1_1080p_256c_b4 refapp_1024x576_128x64x128_b4}; do
  AB_REPS=${AB_REPS:-300} python tools/lib_ab.py $A $B $w V0 V1 V2 VG 2>&1 | grep -v amdgpu.ids
done
