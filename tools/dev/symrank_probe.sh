#!/bin/bash
# dev: every symrank kernel variant built as build/symrank_bench_* (tools/dev/symrank_bench.hip against a copy of
# backend_hip.h) on the recorded block (tools/dev/make_symrank_case.py); optional: "cyc N" = items cycle over N symbols
for b in build/symrank_bench_*; do
  [ -x $b ] || continue
  echo "== $b"; timeout 25 $b build/symrank_case.bin "$@" | tail -1
done
