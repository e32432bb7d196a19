set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
E="python tools/dist_emulate.py"
for pr in 0 -1; do for q in 4 16; do
  echo "## partner priority $pr, GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 600 $E --workload cfg2_1080p_512c_b8 --worlds 8 --batches 1,8 --shares 30 --model 1:2 --comms 0 --partner-priority $pr 2>/dev/null | grep -v "^#"
  GPU_MAX_HW_QUEUES=$q timeout 600 $E --workload cfg3_4k_1024c_b8 --worlds 8 --batches 1 --shares 30 --model 1:2 --comms 0 --partner-priority $pr 2>/dev/null | grep -v "^#"
done; done > $O/emul_priority.txt; cat $O/emul_priority.txt
