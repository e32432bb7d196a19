set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_denoise.py tests/test_ref_gl.py tests/test_benchmark_path.py -m gpu -x -q > gpurun_out/r06/t_5.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06/t_5.log
timeout 600 python bench.py --no-cpu-baseline --pmc off --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(json.dumps(j['present_pass']))"
timeout 600 python bench.py --workload refapp_1024x576_128x64x128_b4 --no-cpu-baseline --pmc off --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(json.dumps(j['present_pass']))"
timeout 600 python bench.py --workload cfg3_4k_1024c_b8 --no-cpu-baseline --pmc off --steps 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(json.dumps(j['present_pass']))"
