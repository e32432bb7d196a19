set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
E="python tools/dist_emulate.py"
# the stand-in with RCCL's two costs (issue order per communicator, a group = one kernel of 2 workgroups per operation), a communicator per slot
timeout 900 $E --workload cfg2_1080p_512c_b8 --worlds 2,4,8 --batches 1,8 --model 1:2 --comms 0 --json $O/emul_headline.json 2>/dev/null > $O/emul_headline.txt; cat $O/emul_headline.txt
# ... every slot on ONE communicator (round 5's pipeline) under the same model, 8 ranks
timeout 600 $E --workload cfg2_1080p_512c_b8 --worlds 8 --batches 1,8 --shares 30 --model 1:2 --comms 1 2>/dev/null | grep -v "^#" > $O/emul_headline_onecomm.txt; cat $O/emul_headline_onecomm.txt
# ... and the copy bandwidth of the modelled kernel: 4 and 8 workgroups per operation (what the link costs is not known before the first run on a node)
for k in 4 8; do timeout 600 $E --workload cfg2_1080p_512c_b8 --worlds 8 --batches 1,8 --shares 30 --model 1:$k --comms 0 2>/dev/null | grep -v "^#"; done > $O/emul_headline_wgs.txt; cat $O/emul_headline_wgs.txt
timeout 900 $E --workload cfg3_4k_1024c_b8 --worlds 2,4,8 --batches 1,8 --model 1:2 --comms 0 --json $O/emul_cfg3.json 2>/dev/null > $O/emul_cfg3.txt; cat $O/emul_cfg3.txt
timeout 600 $E --workload cfg3_4k_1024c_b8 --worlds 8 --batches 1 --shares 30 --model 1:2 --comms 1 2>/dev/null | grep -v "^#" > $O/emul_cfg3_onecomm.txt; cat $O/emul_cfg3_onecomm.txt
timeout 1500 $E --workload cfg4_4k_2048c_b8_sparse --worlds 2,4,8 --batches 1 --shares 100 --model 1:2 --comms 0 --json $O/emul_cfg4.json 2>/dev/null > $O/emul_cfg4.txt; cat $O/emul_cfg4.txt
