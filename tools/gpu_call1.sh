#!/bin/bash
# dev: quick GPU validation of a build: smoke gate, symrank microbench variants, fast-mode tests, bench, kernel stats
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/c1_smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 $OUT/c1_smoke.log; exit 1; }
tail -1 $OUT/c1_smoke.log
timeout 400 python -m pytest tests/test_gpu_fast.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > $OUT/c1_pytest.log
cat $OUT/c1_pytest.log
grep -q passed $OUT/c1_pytest.log && ! grep -q failed $OUT/c1_pytest.log || exit 1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/c1_bench.err | tail -1 > $OUT/c1_bench.json
cat $OUT/c1_bench.json
timeout 120 python tools/dev/members_scale.py 8 2>&1 | tail -1 | tee $OUT/c1_members.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/c1_trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c1_trace_bench.json 2>$OUT/c1_trace.err
DB=$(find $OUT/c1_trace -name '*_results.db' | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $OUT/c1_kernel_stats.csv
rm -rf $OUT/c1_trace
cut -c1-140 $OUT/c1_kernel_stats.csv | head -24
