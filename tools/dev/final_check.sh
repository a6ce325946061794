#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -u -m pytest tests/test_gpu_fast.py -m gpu -x -q --timeout=250 -k "bench_multi_rank or default_mode" 2>&1 | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
