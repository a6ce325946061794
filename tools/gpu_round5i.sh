#!/bin/bash
export TMPDIR=/tmp
python tools/dev/bisect_emu.py fixed-unaligned-loads
timeout 900 python -u -m pytest tests/test_gpu_fast.py tests/test_gpu_verify.py tests/test_gpu_arena.py tests/test_gpu_soak.py -m gpu -x -q --timeout=400 -k "not soak_rotations" > gpurun_out/r05i_pytest.log 2>&1
tail -4 gpurun_out/r05i_pytest.log
