set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -x -q -s -k "real_rccl or communicator or reserve_samples" > gpurun_out/r06/t_new.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r06/t_new.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -5 gpurun_out/r06/t_all.log
timeout 600 python tools/experiments/rccl_slots_probe.py > gpurun_out/r06/rccl_slots.txt 2> gpurun_out/r06/rccl_slots.err; echo "probe rc=$?"; cat gpurun_out/r06/rccl_slots.txt
WLS=cfg2_1080p_512c_b8 timeout 900 bash tools/evidence.sh r06t bench > gpurun_out/r06/evidence_t.log 2>&1; echo "evidence rc=$?"; tail -20 gpurun_out/evidence_r06t/r06t_cfg2_1080p_512c_b8.txt
for m in 0:0 1:2; do for c in 1 0; do timeout 600 python tools/dist_emulate.py --worlds 8 --batches 1,8 --shares 30 --model $m --comms $c 2>/dev/null | grep -v "^#" ; done; done > gpurun_out/r06/emul_headline_quick.txt; cat gpurun_out/r06/emul_headline_quick.txt
