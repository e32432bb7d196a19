set -u
cd $GRAFT_REPO_ROOT
bash tools/evidence.sh r06 > gpurun_out/evidence_r06.log 2>&1; echo "evidence rc=$?"; tail -3 gpurun_out/evidence_r06.log | cut -c1-150
O=gpurun_out/evidence_r06
{ for seed in 6111 6112 6113 6114 6115 6116 6117; do timeout 900 python tools/fuzz_parity.py 1000 $seed 2>&1 | tail -1; done
  for seed in 6211 6212; do timeout 900 python tools/fuzz_parity.py 400 $seed big 2>&1 | tail -1; done
  for seed in 6311 6312; do timeout 900 python tools/fuzz_parity.py 1000 $seed pow2 2>&1 | tail -1; done
  for seed in 6411 6412 6413 6414 6415 6416; do timeout 900 python tools/fuzz_parity.py 1000 $seed pool 2>&1 | tail -1; done; } | grep -v amdgpu >> $O/r06_fuzz.txt
wc -l $O/r06_fuzz.txt; grep -c " 0 mismatching" $O/r06_fuzz.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_final_tests.log 2>&1; echo "gpu tests rc=$?"; grep -a "passed\|failed" gpurun_out/r06_final_tests.log | tail -1
