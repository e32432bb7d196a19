set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/t_final.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r06/t_final.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 300 python bench.py 2>/dev/null | cut -c1-300
