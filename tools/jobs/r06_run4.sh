set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_denoise.py tests/test_reference_parity_gpu.py tests/test_benchmark_path.py -m gpu -x -q > gpurun_out/r06/t_4.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r06/t_4.log
cd /tmp
timeout 600 python $GRAFT_REPO_ROOT/tools/flythrough.py cfg2_1080p_512c_b8 30 --out $GRAFT_REPO_ROOT/gpurun_out/r06/fly_headline.json 2>/dev/null | tail -1
timeout 600 python $GRAFT_REPO_ROOT/tools/flythrough.py refapp_1024x576_128x64x128_b4 30 --out $GRAFT_REPO_ROOT/gpurun_out/r06/fly_refapp.json 2>/dev/null | tail -1
