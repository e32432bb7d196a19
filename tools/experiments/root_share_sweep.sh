#!/bin/bash
# Root share of the multi-GPU pipeline, emulated on one GPU: rank 0 with its peers fed (root_rank.py) against rank 1 rendering
# its share (shard_batch.py), per world size, frames per launch and root share; the frame rate of the job is that of the
# slower of the two.   usage: root_share_sweep.sh [frames per launch] [worlds] [shares]
cd ${GRAFT_REPO_ROOT:-.}
BATCH=${1:-4}; WORLDS=${2:-"2 4 8"}; SHARES=${3:-"0,85,70,55,40,30"}
for world in $WORLDS; do
  python tools/experiments/root_rank.py $world $BATCH 4 $SHARES 2>&1 | grep "rank 0 of"
  for wgt in ${SHARES//,/ }; do python tools/experiments/shard_batch.py $world $wgt 2>&1 | grep "$BATCH frame(s) per launch, 4 launches"; done
done
