#!/bin/bash
export TMPDIR=/tmp
echo "== hints checked against the device"
ORZ_CHECK_HINTS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members > /tmp/b.json 2>/tmp/b.err; tail -2 /tmp/b.err
python -c "
import json; d=json.load(open('/tmp/b.json')); print({k:d[k] for k in ('value','ms_per_step','roundtrip_ok','compressed_bytes','host_syncs_per_block','host_syncs_per_unit')})"
echo "== normal"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-members > /tmp/b.json 2>/tmp/b.err; tail -2 /tmp/b.err
python -c "
import json; d=json.load(open('/tmp/b.json')); print({k:d[k] for k in ('value','ms_per_step','roundtrip_ok','compressed_bytes','host_syncs_per_block','host_syncs_per_unit')})"
timeout 1500 python -u -m pytest tests -m gpu -q --timeout=500 > gpurun_out/r06s_pytest.log 2>&1; tail -8 gpurun_out/r06s_pytest.log
