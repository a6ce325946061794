#!/bin/bash
# dev: soak of the symbol-ranking guard (see soak_members.py); the guard's messages are counted from stderr
OUT=gpurun_out
mkdir -p $OUT
timeout ${2:-400} python tools/dev/soak_members.py ${1:-150} 8 > $OUT/r03_soak.json 2> $OUT/r03_soak.err
echo "{\"guard_messages\": $(grep -c 'was repeated' $OUT/r03_soak.err), \"two_runs_differ_messages\": $(grep -c 'from the same tables differ' $OUT/r03_soak.err), \"soak_failures\": $(grep -c '^SOAK' $OUT/r03_soak.err), \"env\": \"ORZ_SYMRANK_VERIFY=${ORZ_SYMRANK_VERIFY:-0} ORZ_FAST_VERIFY=${ORZ_FAST_VERIFY:-0}\"}" >> $OUT/r03_soak.json
cat $OUT/r03_soak.json; grep -v "^Traceback\|^  " $OUT/r03_soak.err | head -12
