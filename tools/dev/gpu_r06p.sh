#!/bin/bash
# dev: the lead block cut into units (ORZ_FAST_LEADUNIT) / every block (ORZ_FAST_UNIT), now that ranking keeps up with the parse
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-members > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'P'
import json,sys
d=json.load(open('/tmp/b.json'))
print(json.dumps({"env":sys.argv[1], **{k:d[k] for k in ('value','ms_per_step','size_delta_pct','roundtrip_ok','compressed_bytes','host_syncs_per_block')}}))
P
}
{
run X=0
run ORZ_FAST_LEADUNIT=2097152
run ORZ_FAST_LEADUNIT=4194304
run ORZ_FAST_LEADUNIT=8388608
run ORZ_FAST_UNIT=8388608
run ORZ_FAST_UNIT=8388608 ORZ_FAST_LEADUNIT=2097152
run ORZ_FAST_UNIT=4194304
} | tee gpurun_out/r06p_units.jsonl
