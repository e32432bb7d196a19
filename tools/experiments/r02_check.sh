#!/bin/bash
# Runs on the GPU box (via gpurun): the GPU test suite, the live oracle/_ref tests, smoke and a default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02_check}
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python -m pytest tests/test_ref_gl.py -q -m "not gpu" > $OUT/pytest_ref_live.log 2>&1; echo "ref live rc=$?"; tail -3 $OUT/pytest_ref_live.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
r = d["roofline"]
print({k: d[k] for k in ("metric", "value", "ms_per_step", "ms_per_step_single_stream")})
print({k: r[k] for k in ("achieved", "frac", "traffic", "kernel_ms_avg", "issued_bytes", "issued_GBps", "issue_ipc")})
print(r["traffic_note"]); print(r["kernel_ms_per_view"]); print(d["config"]["primary_hit_fraction"])
print(d.get("cpu_baseline")); print(d.get("cpu_baseline_port"))
PY
