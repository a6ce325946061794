#!/bin/bash
# Race check of the encoder's kernels (see race_main.cpp): builds the threaded emulation with ThreadSanitizer and runs it on the
# given inputs.  Exit code 0 = no data race reported and every stream was written.
#   tests/race/run.sh [-t THREADS] [-s] <fast|exact> <input file> [...]      -s: the self-test build (round 3's racy word repair)
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
T=8; SELF=""
while getopts "t:s" o; do case $o in t) T=$OPTARG;; s) SELF="-DORZ_RACE_SELFTEST";; esac; done
shift $((OPTIND - 1))
MODE=${1:?mode}; shift
OUT=${RACE_OUT:-/tmp/orz_race}; mkdir -p "$OUT"
BIN="$OUT/race_main_$T$( [ -n "$SELF" ] && echo _selftest )"
SRCS="$HERE/race_main.cpp $HERE/../emu/emu_backend.cpp $HERE/../emu/simt.h $HERE/../../orz_amd/csrc/*.h"
if [ ! -x "$BIN" ] || [ -n "$(find $SRCS -newer "$BIN" 2>/dev/null)" ]; then
  g++ -O1 -g -std=c++17 -fsanitize=thread -DORZ_EMU_THREADS=$T $SELF -Wno-unknown-pragmas -o "$BIN" "$HERE/race_main.cpp" -lpthread || exit 2
fi
rc=0
for f in "$@"; do
  log="$OUT/$(basename "$f").$MODE$( [ -n "$SELF" ] && echo .selftest ).log"
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 exitcode=0" "$BIN" "$MODE" "$f" > "$log" 2>&1 || rc=1
  n=$(grep -c "WARNING: ThreadSanitizer: data race" "$log")
  echo "$f ($MODE): $n data races reported; $(tail -n 1 "$log" | cut -c1-120)"
  grep "SUMMARY" "$log" | sort | uniq -c | sort -rn | head -8
  [ "$n" != "0" ] && rc=1
done
exit $rc
