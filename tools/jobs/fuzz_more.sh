cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
{ for seed in 6121 6122 6123 6124 6125 6126; do timeout 900 python tools/fuzz_parity.py 1000 $seed 2>&1 | tail -1; done
  for seed in 6221 6222; do timeout 900 python tools/fuzz_parity.py 400 $seed big 2>&1 | tail -1; done
  for seed in 6321 6322; do timeout 900 python tools/fuzz_parity.py 1000 $seed pow2 2>&1 | tail -1; done
  for seed in 6421 6422 6423 6424 6425; do timeout 900 python tools/fuzz_parity.py 1000 $seed pool 2>&1 | tail -1; done; } | grep -v amdgpu > gpurun_out/r06/fuzz_more.txt
wc -l gpurun_out/r06/fuzz_more.txt; grep -c " 0 mismatching" gpurun_out/r06/fuzz_more.txt
