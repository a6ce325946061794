#!/bin/bash
# dev: symrank kernel variants on the recorded block (build/symrank_bench_*), real stream and an all-ranks<64 synthetic one
for b in build/symrank_bench_old build/symrank_bench_trips*; do
  [ -x $b ] || continue
  echo "== $b"; timeout 25 $b build/symrank_case.bin | tail -1
  echo "   cyc60:"; timeout 25 $b build/symrank_case.bin cyc 60 | tail -1
done
