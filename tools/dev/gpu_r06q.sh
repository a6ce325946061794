#!/bin/bash
# dev: unit sizes for the single stream, and what 8 MiB units do to eight encoders
export TMPDIR=/tmp
run() {
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-members > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'P'
import json,sys
d=json.load(open('/tmp/b.json'))
print(json.dumps({"env":sys.argv[1], **{k:d[k] for k in ('value','ms_per_step','roundtrip_ok','compressed_bytes','host_syncs_per_block')}}))
P
}
{
run ORZ_FAST_UNIT=6291456
run ORZ_FAST_UNIT=8388608
run ORZ_FAST_UNIT=10485760
run ORZ_FAST_UNIT=12582912
for u in 16777216 8388608; do
  ORZ_FAST_UNIT=$u timeout 300 python tools/dev/members_scale.py 1 8 8 2>/dev/null | sed "s/^/unit $u: /"
done
} | tee gpurun_out/r06q_units.jsonl
