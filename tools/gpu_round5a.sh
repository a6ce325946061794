#!/bin/bash
# Round 5, first call: the list form of the repair stage on the hardware -- GPU == emulation (golden hashes), a slice of the fast-mode
# tier, the bench line with its members annex, the grid form beside it, kernel stats of both, members aggregate of both.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 420 python -u -m pytest tests/test_gpu_soak.py tests/test_gpu_fast.py -m gpu -x -q --timeout=300 \
  -k "emulation or reproducer or round_trip_and_size or two_blocks or valid_plan or host_emulation" > $OUT/r05a_pytest.log 2>&1
tail -4 $OUT/r05a_pytest.log
timeout 400 python bench.py --steps 3 --warmup 1 > $OUT/r05a_bench.json 2> $OUT/r05a_bench.err
tail -c 2500 $OUT/r05a_bench.json
ORZ_FAST_REPAIR=grid timeout 200 python bench.py --steps 3 --warmup 1 --no-members --no-cpu-baseline > $OUT/r05a_bench_grid.json 2>> $OUT/r05a_bench.err
python - <<PY
import json
for f in ("r05a_bench.json", "r05a_bench_grid.json"):
    try:
        d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ["value", "ms_per_step", "compressed_bytes", "roundtrip_ok", "size_delta_pct", "stage_seconds_per_step"]})
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp
for form in lists grid; do
  [ $form = grid ] && export ORZ_FAST_REPAIR=grid
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/../gpurun_out/r05a_trace_$form -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members > /dev/null 2>$REPO/$OUT/r05a_trace_$form.err
  DB=$(find $REPO/$OUT/r05a_trace_$form -name '*_results.db' | head -1)
  [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $REPO/$OUT/r05a_kernel_stats_$form.csv
  rm -rf $REPO/$OUT/r05a_trace_$form
done
unset ORZ_FAST_REPAIR
cd $REPO
head -30 $OUT/r05a_kernel_stats_lists.csv | cut -c1-140
timeout 150 python tools/dev/members_scale.py 1 8 8 > $OUT/r05a_members_lists.jsonl 2>$OUT/r05a_members.err
ORZ_FAST_REPAIR=grid timeout 100 python tools/dev/members_scale.py 8 > $OUT/r05a_members_grid.jsonl 2>>$OUT/r05a_members.err
cat $OUT/r05a_members_lists.jsonl $OUT/r05a_members_grid.jsonl
