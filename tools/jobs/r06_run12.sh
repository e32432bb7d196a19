cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 300 python tools/experiments/upload_rate.py 2>/dev/null | tee gpurun_out/r06/pcie.txt
