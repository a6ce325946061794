#!/bin/bash
# Round 5, second call: kernel stats of the bench workload, repair stage as lists (ordinals by ballots / by the LDS table) and as grids
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for form in lists table grid; do
  unset ORZ_FAST_REPAIR ORZ_FAST_ORD
  [ $form = grid ] && export ORZ_FAST_REPAIR=grid
  [ $form = table ] && export ORZ_FAST_ORD=table
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r05b_trace_$form -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members > $OUT/r05b_bench_$form.json 2>$OUT/r05b_trace_$form.err
  DB=$(find $OUT/r05b_trace_$form -name '*_results.db' | head -1)
  [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $OUT/r05b_kernel_stats_$form.csv
  rm -rf $OUT/r05b_trace_$form
done
cd $REPO
head -12 $OUT/r05b_kernel_stats_lists.csv | cut -c1-140
