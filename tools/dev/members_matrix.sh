#!/bin/bash
# dev: aggregate rate of 4 / 8 encoders on one GPU under hardware-queue counts and with / without the round loop's hipGraph
#   bash tools/dev/members_matrix.sh <tag>
TAG=${1:-mm}
OUT=gpurun_out
mkdir -p $OUT
for rep in 1 2; do
  for combo in "16 1" "16 0" "32 1" "32 0" "8 1"; do
    set -- $combo
    echo "{\"hw_queues\": $1, \"graphs\": $2, \"rep\": $rep}" >> $OUT/${TAG}_members_matrix.jsonl
    GPU_MAX_HW_QUEUES=$1 ORZ_GRAPHS=$2 timeout 300 python tools/dev/members_scale.py 4 8 >> $OUT/${TAG}_members_matrix.jsonl 2>>$OUT/${TAG}_members_matrix.err
  done
done
cat $OUT/${TAG}_members_matrix.jsonl
