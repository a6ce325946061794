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
for u in 4194304 16777216; do echo unit $u; ORZ_FAST_UNIT=$u timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b['compressed_bytes'], b['stage_seconds_per_step'])"; done 2>&1 | tee $OUT/c1_units.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/c1_trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c1_trace_bench.json 2>$OUT/c1_trace.err
DB=$(find $OUT/c1_trace -name '*_results.db' | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $OUT/c1_kernel_stats.csv
rm -rf $OUT/c1_trace
cut -c1-140 $OUT/c1_kernel_stats.csv | head -24
