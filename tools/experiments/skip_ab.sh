#!/bin/bash
# A/B of the skip to the occupied-cell box (tuning flag VRT_TUNE_NO_SKIP_TO_BOX = 0x01): parity tests + fuzz with it on, kernel time per view both ways.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-skip_ab}
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_ref_gl.py tests/test_benchmark_path.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
python tools/fuzz_parity.py 400 4101 > $OUT/fuzz_small.log 2>&1; echo "fuzz small rc=$?"; tail -3 $OUT/fuzz_small.log
python tools/fuzz_parity.py 200 4102 big > $OUT/fuzz_big.log 2>&1; echo "fuzz big rc=$?"; tail -3 $OUT/fuzz_big.log
for wl in cfg2_1080p_512c_b8 cfg2_1080p_512c_b4 cfg3_4k_1024c_b8; do
  echo "== $wl: with the jump (flags 0x00) / every cell walked (flags 0x01)"; python tools/variant_sweep.py $wl 0,0/0x01 200 2>&1 | tail -2
done | tee $OUT/sweep.log
