#!/bin/bash
# dev: the whole GPU tier once more on the shipped library
export TMPDIR=/tmp
timeout 1100 python -u -m pytest tests -m gpu -q --timeout=400 --durations=8 > gpurun_out/r05_pytest_gpu.log 2>&1
tail -12 gpurun_out/r05_pytest_gpu.log
