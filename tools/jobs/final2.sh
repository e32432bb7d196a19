set -u
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_dist_fake_rccl.py -m gpu -q -x --timeout=120 -k "kept_in_the_pool or a_communicator_per_launch_slot" 2>&1 | tail -1; done
timeout 900 python -m pytest tests -m gpu -x -q --timeout=300 > gpurun_out/r06_final_tests.log 2>&1; echo "gpu tests rc=$?"; grep -a "passed\|failed" gpurun_out/r06_final_tests.log | tail -1
rm -rf gpurun_out/evidence_r06
bash tools/evidence.sh r06 > gpurun_out/evidence_r06.log 2>&1; echo "evidence rc=$?"; tail -3 gpurun_out/evidence_r06.log | cut -c1-150
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
