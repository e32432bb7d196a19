set -u
cd ${GRAFT_REPO_ROOT:-.}
python tools/variant_sweep.py cfg2_1080p_512c_b8 0 300 V0,V1,V2 2>&1 | grep -v amdgpu.ids | tail -1
python bench.py --no-cpu-baseline > gpurun_out/qb.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/qb.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["ms_per_step_single_stream"], r["kernel_ms_avg"], r["insts_per_launch"]["SQ_INSTS_VALU"], r["insts_per_launch"]["SQ_INSTS_SALU"], r["kernel_ms_per_view"])
PY
