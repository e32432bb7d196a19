#!/bin/bash
# Runs on the GPU box (via gpurun): parity suite, then the bench line twice (short form).
# usage: tools/experiments/gpu_check.sh [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
for i in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.0f  kernel_ms %.4f  frac %.3f ' % (d['value'], r['kernel_ms_avg'], r['frac']), r['kernel_ms_per_view'])"
done
