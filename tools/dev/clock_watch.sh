#!/bin/bash
# dev: shader / memory clocks the GPU runs at while bench.py encodes (sampled with rocm-smi), then once more after asking
# for the high performance level -- is the ranking chain's time per item a clock question?
OUT=gpurun_out
mkdir -p $OUT
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -20 > $OUT/clock_idle.txt
sample() {
  for i in $(seq 1 400); do rocm-smi -c 2>/dev/null | grep -E "sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.05; done
}
run() {
  sample > $OUT/clock_$1.txt &
  SP=$!
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'run': '$1', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_us': d['roofline']['avg_launch_us']}))" | tee -a $OUT/clock_bench.jsonl
  kill $SP 2>/dev/null
  wait $SP 2>/dev/null
}
run auto
rocm-smi --setperflevel high > $OUT/clock_set.txt 2>&1
run high
rocm-smi --setperflevel auto >> $OUT/clock_set.txt 2>&1
for f in auto high; do echo "== $f"; sort $OUT/clock_$f.txt | uniq -c | sort -rn | head -6; done
cat $OUT/clock_idle.txt $OUT/clock_set.txt | head -40
