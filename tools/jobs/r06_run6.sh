set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python tools/experiments/bounce_lanes.py 2>/dev/null > gpurun_out/r06/bounce_lanes.txt; cat gpurun_out/r06/bounce_lanes.txt
timeout 300 python tools/experiments/bounce_cost.py 2>/dev/null > gpurun_out/r06/bounce_cost.txt; cat gpurun_out/r06/bounce_cost.txt
